"""-m gpu: plain StrongSORT on the GPU (tlk_ssort_* through the C ABI) against the reference's golden runs and the oracle."""
import json
import os

import numpy as np
import pytest

from conftest import GOLDEN
from test_oracle_ssort import RUNS, replay

pytestmark = pytest.mark.gpu


class GpuTracker:
    def __init__(self, D, hp, **kw):
        from tracklab_amd._lib import SsortBank
        self.bank = SsortBank(D, **hp, **kw)

    def update(self, dets, emb):
        r = self.bank.update(dets, emb)
        return np.column_stack([r["ltrb"], r["track_id"], r["class_id"], r["conf"], r["det_id"]]).astype(np.float64).reshape(-1, 8)

    def tracks(self):
        return self.bank.tracks()


def check_state_gpu(trk, g, f):
    ids, mean, cov, feat, st, gl = trk.tracks()
    np.testing.assert_array_equal(ids, g[f"f{f}_track_ids"])
    np.testing.assert_array_equal(st, g[f"f{f}_state"])
    np.testing.assert_array_equal(gl, g[f"f{f}_gallery"])
    np.testing.assert_array_equal(mean, g[f"f{f}_mean"])                     # library operation order (oracle/src/lapack_order.h): bit-exact
    np.testing.assert_array_equal(cov, g[f"f{f}_cov"])
    np.testing.assert_allclose(feat, g[f"f{f}_feat"], rtol=0, atol=5e-7)        # float32 EMA + renorm: summation order of the norms


@pytest.mark.parametrize("name", RUNS)
def test_plain_strongsort_gpu_matches_reference(name):
    replay(name, lambda D, hp: GpuTracker(D, hp), check_state_gpu)


def test_plain_strongsort_gpu_kf_state_bit_exact_vs_oracle(orc):
    """Same inputs to oracle and GPU: ids/boxes identical, KF means/covariances bit-identical (same fp64 op order)."""
    from tracklab_amd.synth import SyntheticStream
    hp = dict(max_dist=0.2, max_iou_dist=0.7, max_age=15, max_unmatched_preds=7, n_init=3, nn_budget=20, mc_lambda=0.995, ema_alpha=0.9)
    D = 64
    gpu, cpu = GpuTracker(D, hp), orc.PlainStrongSORT(D, **hp)
    for fr in SyntheticStream(11, 40, 90, parts=1, dim=D, with_embeddings=True, miss_prob=0.08, churn_period=20):
        dets, emb = fr["dets"], fr["embeddings"][:, 0, :].astype(np.float32)
        a, b = gpu.update(dets, emb), cpu.update(dets, emb)
        np.testing.assert_array_equal(a, b)
        gi, gm, gc, gf, gs, gg = gpu.tracks()
        ci, cm, cc, cf, cs, cg = cpu.tracks()
        np.testing.assert_array_equal(gi, ci); np.testing.assert_array_equal(gs, cs); np.testing.assert_array_equal(gg, cg)
        np.testing.assert_array_equal(gm, cm); np.testing.assert_array_equal(gc, cc)
        np.testing.assert_allclose(gf, cf, rtol=0, atol=5e-7)


def test_plain_strongsort_bank_streams_are_independent_and_batched_frames_match():
    """update_dev over (streams, frames) == per-stream host updates; reset(stream) restarts ids at 1."""
    import torch
    from tracklab_amd._lib import SSORT_ROW, SsortBank
    from tracklab_amd.synth import SyntheticStream
    D, S, F, MAXD = 32, 3, 12, 64
    hp = dict(max_age=10, nn_budget=8)
    bank = SsortBank(D, **hp, n_streams=S, max_dets=MAXD, max_tracks=128)
    ref = [SsortBank(D, **hp, max_dets=MAXD, max_tracks=128) for _ in range(S)]
    dets = np.zeros((S, F, MAXD, 7)); feat = np.zeros((S, F, MAXD, D), np.float32); counts = np.zeros((S, F), np.int32)
    expect = [[None] * F for _ in range(S)]
    for s in range(S):
        for f, fr in enumerate(SyntheticStream(20 + s, 25, F, parts=1, dim=D, with_embeddings=True)):
            n = len(fr["dets"])
            dets[s, f, :n] = fr["dets"]; feat[s, f, :n] = fr["embeddings"][:, 0, :]; counts[s, f] = n
            expect[s][f] = ref[s].update(fr["dets"], fr["embeddings"][:, 0, :])
    d_dets, d_feat, d_cnt = torch.from_numpy(dets).cuda(), torch.from_numpy(feat).cuda(), torch.from_numpy(counts).cuda()
    cap = 128
    rows = torch.zeros((S, F, cap, SSORT_ROW.itemsize), dtype=torch.uint8, device="cuda")
    ocnt = torch.zeros((S, F), dtype=torch.int32, device="cuda")
    bank.update_dev(d_dets.data_ptr(), d_feat.data_ptr(), d_cnt.data_ptr(), F, rows.data_ptr(), cap, ocnt.data_ptr())
    torch.cuda.synchronize()
    got = rows.cpu().numpy().view(SSORT_ROW).reshape(S, F, cap)
    oc = ocnt.cpu().numpy()
    for s in range(S):
        for f in range(F):
            e = expect[s][f]
            assert oc[s, f] == len(e)
            for name in SSORT_ROW.names:
                np.testing.assert_array_equal(got[s, f, :len(e)][name], e[name])
    bank.reset(1)
    fr = next(iter(SyntheticStream(99, 5, 1, parts=1, dim=D, with_embeddings=True)))
    for _ in range(4):
        r = bank.update(fr["dets"], fr["embeddings"][:, 0, :], stream=1)
    assert sorted(r["track_id"]) == list(range(1, len(fr["dets"]) + 1))


def test_plain_strongsort_rejects_bad_configuration():
    from tracklab_amd._lib import SsortBank, TlkError
    with pytest.raises(TlkError):
        SsortBank(48)                       # dim not supported by the MFMA tile
    with pytest.raises(TlkError):
        SsortBank(64, nn_budget=0)
    b = SsortBank(64, max_dets=8)
    with pytest.raises(TlkError):
        b.update(np.zeros((9, 7)), np.ones((9, 64), np.float32))


def test_plain_strongsort_beyond_512_tracks(orc):
    """max_tracks up to 1024: three 230-object scenes in turn (n_init 1, max_age 100) leave ~690 tracks -- more rows than the register-resident
    Hungarian solver holds; rows and Kalman state stay identical to the oracle, the first scene is re-identified when it returns."""
    from tracklab_amd.synth import SyntheticStream
    hp = dict(max_dist=0.2, max_iou_dist=0.7, max_age=100, max_unmatched_preds=7, n_init=1, nn_budget=4, mc_lambda=0.995, ema_alpha=0.9)
    D = 32
    gpu, cpu = GpuTracker(D, hp, max_tracks=1024, max_dets=256), orc.PlainStrongSORT(D, **hp)
    scenes = [iter(SyntheticStream(80 + k, 230, 8, parts=1, dim=D, with_embeddings=True, miss_prob=0.1)) for k in range(3)]
    most = 0
    for f, k in enumerate([0, 0, 0, 1, 1, 1, 2, 2, 2, 0, 0, 1, 1, 2, 2]):
        fr = next(scenes[k])
        dets, emb = fr["dets"].copy(), fr["embeddings"][:, 0, :].astype(np.float32)
        dets[:, 6] += 100000 * k
        a, b = gpu.update(dets, emb), cpu.update(dets, emb)
        np.testing.assert_array_equal(a, b, err_msg=f"frame {f}")
        gi, gm, gc, _, gs, gg = gpu.tracks()
        ci, cm, cc, _, cs, cg = cpu.tracks()
        np.testing.assert_array_equal(gi, ci); np.testing.assert_array_equal(gs, cs); np.testing.assert_array_equal(gg, cg)
        np.testing.assert_array_equal(gm, cm); np.testing.assert_array_equal(gc, cc)
        most = max(most, len(gi))
    assert most > 540


def test_plain_strongsort_unbounded_gallery(orc):
    """nn_budget=None (the reference keeps every sample, nn_matching.py:124-142): same rows, state and gallery lengths as the oracle while the
    galleries fit the rows reserved per track; a track that outgrows them is a loud error, never a silently dropped sample."""
    from tracklab_amd._lib import TlkError
    from tracklab_amd.synth import SyntheticStream
    hp = dict(max_dist=0.2, max_iou_dist=0.7, max_age=15, max_unmatched_preds=7, n_init=1, nn_budget=None, mc_lambda=0.995, ema_alpha=0.9)
    D = 64
    gpu, cpu = GpuTracker(D, hp, gallery_rows=64), orc.PlainStrongSORT(D, **hp)
    small = GpuTracker(D, hp, gallery_rows=5)
    failed_at = None
    for fr in SyntheticStream(12, 20, 30, parts=1, dim=D, with_embeddings=True, miss_prob=0.05):
        dets, emb = fr["dets"], fr["embeddings"][:, 0, :].astype(np.float32)
        a, b = gpu.update(dets, emb), cpu.update(dets, emb)
        np.testing.assert_array_equal(a, b)
        gi, gm, gc, _, gs, gg = gpu.tracks()
        ci, cm, cc, _, cs, cg = cpu.tracks()
        np.testing.assert_array_equal(gi, ci); np.testing.assert_array_equal(gg, cg); np.testing.assert_array_equal(gm, cm)
        if failed_at is None:
            try:
                small.update(dets, emb)
            except TlkError:
                failed_at = fr["frame"]
    assert gg.max() > 20                                   # far beyond any of the budgets the other tests use: nothing was dropped
    assert failed_at is not None and 4 <= failed_at <= 8   # the sixth sample of a track does not fit 5 rows


def test_plain_strongsort_1000_tracks_300_detections(orc):
    """Capacity is an allocation size (r04; the reference's track list grows, strong_sort/sort/tracker.py:130-141): 300-object scenes shown in turn
    (n_init 1, max_age 100) leave over 1000 tracks beside 300 detections per frame -- past both LDS tiers, lists and Hungarian work area in
    HBM -- rows and Kalman state equal the oracle every frame; a small scene afterwards runs in the LDS tier again."""
    from tracklab_amd.synth import SyntheticStream
    hp = dict(max_dist=0.2, max_iou_dist=0.7, max_age=100, max_unmatched_preds=7, n_init=1, nn_budget=4, mc_lambda=0.995, ema_alpha=0.9)
    D = 32
    gpu, cpu = GpuTracker(D, hp, max_tracks=4096, max_dets=512), orc.PlainStrongSORT(D, **hp)
    scenes = [iter(SyntheticStream(90 + k, 300, 4, parts=1, dim=D, with_embeddings=True, miss_prob=0.05)) for k in range(5)]
    small = iter(SyntheticStream(77, 20, 3, parts=1, dim=D, with_embeddings=True))
    most = 0
    for f, k in enumerate([0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 0, -1, -1, 1]):
        fr = next(small) if k < 0 else next(scenes[k])
        dets, emb = fr["dets"].copy(), fr["embeddings"][:, 0, :].astype(np.float32)
        dets[:, 6] += 100000 * (k + 1)
        a, b = gpu.update(dets, emb), cpu.update(dets, emb)
        np.testing.assert_array_equal(a, b, err_msg=f"frame {f}")
        gi, gm, gc, _, gs, gg = gpu.tracks()
        ci, cm, cc, _, cs, cg = cpu.tracks()
        np.testing.assert_array_equal(gi, ci); np.testing.assert_array_equal(gs, cs); np.testing.assert_array_equal(gg, cg)
        np.testing.assert_array_equal(gm, cm); np.testing.assert_array_equal(gc, cc)
        most = max(most, len(gi))
    assert most > 1000, most          # tracks + 300 detections: beyond the 1024 x 256 LDS tier
