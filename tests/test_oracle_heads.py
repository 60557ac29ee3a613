"""CPU: the oracle's restatement of the r06 prediction heads (oracle/src/heads.c) against a float64 numpy / torch statement of the same heads --
the third-party graphs they stand for (rtmlib's YOLOX ONNX head, the torchreid fork's part-based head) are absent from the reference tree, so
parity with those stays unpinned; what is pinned here is that the checker computes the head the modules define."""
import numpy as np


def test_oracle_yolox_head_is_the_decoupled_head(orc):
    rng = np.random.default_rng(0)
    B, C, ncls = 2, 48, 2
    sizes = [(6, 5), (3, 3)]
    cfs = [rng.standard_normal((B, h * w, C)).astype(np.float32) for h, w in sizes]
    rfs = [rng.standard_normal((B, h * w, C)).astype(np.float32) for h, w in sizes]
    ws = [(rng.standard_normal((5 + ncls, C)) * 0.2).astype(np.float32) for _ in sizes]
    bs = [rng.standard_normal(5 + ncls).astype(np.float32) for _ in sizes]
    got = orc.yolox_head(cfs, rfs, ws, bs, ncls)
    off = 0
    for cf, rf, w, b in zip(cfs, rfs, ws, bs):
        reg = rf.astype(np.float64) @ w[:5].T + b[:5]
        cls = cf.astype(np.float64) @ w[5:].T + b[5:]
        ref = np.concatenate([reg[..., :4], 1 / (1 + np.exp(-reg[..., 4:5])), 1 / (1 + np.exp(-cls))], -1)
        np.testing.assert_allclose(got[:, off:off + cf.shape[1]], ref, rtol=1e-5, atol=1e-5)
        off += cf.shape[1]
    assert off == got.shape[1]


def test_oracle_reid_part_head_is_the_torch_head(orc):
    import torch
    rng = np.random.default_rng(1)
    N, hw, D, K = 5, 24, 32, 6
    f = rng.standard_normal((N, hw, D)).astype(np.float32)
    w, b = (rng.standard_normal((K, D)) * 0.3).astype(np.float32), rng.standard_normal(K).astype(np.float32)
    emb, vis, bad = orc.reid_part_head(f, w, b, 0.5 / K)
    ft = torch.from_numpy(f).double()
    att = torch.softmax(ft @ torch.from_numpy(w).double().T + torch.from_numpy(b).double(), dim=-1)      # (N, hw, K): PartBasedReID.head off the GPU route
    ref = torch.einsum("npk,npd->nkd", att, ft) / att.sum(1).clamp_min(1e-6)[..., None]
    rvis = att.amax(1) > 0.5 / K
    rvis[:, 0] = True
    assert not bad
    np.testing.assert_allclose(emb, ref.numpy(), rtol=1e-5, atol=1e-6)
    np.testing.assert_array_equal(vis.astype(bool), rvis.numpy())
    # hand-off layout: dense batch, padding rows zero, garbage beyond the live rows never read
    counts, maxd = np.array([2, 0, 3], np.int32), 4
    base = np.array([0, 2, 2], np.int32)
    f2 = f.copy()
    f2[5:] = np.nan
    e2, v2, bad2 = orc.reid_part_head(f2, w, b, 0.5 / K, counts, base, maxd)
    assert not bad2 and e2.shape == (12, K, D)
    np.testing.assert_array_equal(e2.reshape(3, maxd, K, D)[0, :2], emb[:2])
    np.testing.assert_array_equal(e2.reshape(3, maxd, K, D)[2, :3], emb[2:5])
    assert np.all(e2.reshape(3, maxd, K, D)[1] == 0) and np.all(e2.reshape(3, maxd, K, D)[0, 2:] == 0) and np.all(v2.reshape(3, maxd, K)[1] == 0)
    f2[1, 3, 4] = np.inf
    assert orc.reid_part_head(f2, w, b, 0.5 / K, counts, base, maxd)[2]
