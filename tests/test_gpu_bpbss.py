"""-m gpu: HIP BPBReID-StrongSORT (libtlk through the C ABI) vs the reference golden vectors and vs the C oracle on
fresh streams; MFMA part-distance kernel vs the oracle. ids / indices / counters / states exact, fp64 boxes 1e-7,
fp32-derived ReID distances 1e-5."""
import glob
import json
import os

import numpy as np
import pytest

from conftest import GOLDEN
from test_oracle_bpbss import bpbss_inputs, check_bpbss_rows

pytestmark = pytest.mark.gpu
BPB_FILES = sorted(glob.glob(os.path.join(GOLDEN, "bpbss_*.npz")))


def _bank(cfg, K, D, **kw):
    from tracklab_amd._lib import BpbssBank
    return BpbssBank(K, D, **cfg, **kw)


@pytest.mark.parametrize("path", BPB_FILES, ids=[os.path.basename(p)[6:-4] for p in BPB_FILES])
def test_hip_bpbss_matches_reference_golden(path):
    g = np.load(path)
    cfg = json.loads(str(g["config"]))
    K, D = int(g["parts"]), int(g["dim"])
    bank = _bank(cfg, K, D, wrapper_mode=True)
    for f, (ids, ltwh, emb, vis, conf, kp) in enumerate(bpbss_inputs(g)):
        rows = bank.update(ids, ltwh, emb, vis, conf, keypoints=kp)
        check_bpbss_rows(g, f, rows)
        if f"f{f}_track_ids" in g:
            tid, mean, cov, feat, fvis = bank.tracks()
            np.testing.assert_array_equal(tid, g[f"f{f}_track_ids"])
            np.testing.assert_array_equal(mean, g[f"f{f}_mean"])                       # library operation order (oracle/src/lapack_order.h): bit-exact
            np.testing.assert_array_equal(cov, g[f"f{f}_cov"])
            np.testing.assert_array_equal(feat, g[f"f{f}_feat"])
            np.testing.assert_array_equal(fvis.astype(bool), g[f"f{f}_fvis"].astype(bool))
    bank.close()


YAML = dict(ema_alpha=0.9, mc_lambda=0.995, max_dist=0.5, motion_criterium="iou", max_iou_distance=0.8,
            max_oks_distance=0.7, max_age=300, n_init=0, nn_budget=100, min_bbox_confidence=0.0,
            only_position_for_kf_gating=False, max_kalman_prediction_without_update=7,
            matching_strategy="strong_sort_matching", gating_thres_factor=1, w_kfgd=1, w_reid=1, w_st=1)


@pytest.mark.parametrize("seed,nobj,D,kw", [(31, 100, 256, {}), (32, 40, 64, {"miss_prob": 0.25, "churn_period": 6})])
def test_hip_bpbss_matches_oracle_fresh_streams(orc, seed, nobj, D, kw):
    from tracklab_amd.synth import SyntheticStream, ltrb_to_ltwh_rows
    K = 6
    bank = _bank(YAML, K, D)
    ref = orc.StrongSORT(K, D, **YAML)
    for fr in SyntheticStream(seed, nobj, 60, parts=K, dim=D, with_embeddings=True, **kw):
        d = fr["dets"]
        ltwh, conf, ids = ltrb_to_ltwh_rows(d[:, :4]), d[:, 4], d[:, 6].astype(np.int64)
        exp = ref.update(ids, ltwh, fr["embeddings"], fr["visibility"], conf)
        got = bank.update(ids, ltwh, fr["embeddings"], fr["visibility"], conf)
        assert len(got) == len(exp)
        for name in ("det_id", "track_id", "hits", "age", "tsu", "state", "matched_name", "pred_valid"):
            np.testing.assert_array_equal(got[name], exp[name], err_msg=name)
        np.testing.assert_array_equal(got["kf_ltwh"], exp["kf_ltwh"])
        np.testing.assert_allclose(got["matched_dist"], exp["matched_dist"], rtol=1e-5, atol=1e-5)
    bank.close()


def test_hip_bpbss_device_batched_entry_point(orc):
    import torch
    from tracklab_amd._lib import BPBSS_ROW
    from tracklab_amd.synth import SyntheticStream, ltrb_to_ltwh_rows
    S, F, MAXD, K, D = 2, 20, 64, 6, 32
    bank = _bank(YAML, K, D, n_streams=S, max_dets=MAXD, wrapper_mode=True)
    ids = np.zeros((S, F, MAXD), dtype=np.int64)
    ltwh = np.zeros((S, F, MAXD, 4))
    ltwh[..., 2:] = 1
    emb = np.zeros((S, F, MAXD, K, D), dtype=np.float32)
    vis = np.zeros((S, F, MAXD, K), dtype=np.uint8)
    conf = np.zeros((S, F, MAXD))
    counts = np.zeros((S, F), dtype=np.int32)
    exp = [[None] * F for _ in range(S)]
    for s in range(S):
        ref = orc.StrongSORT(K, D, **YAML)
        for f, fr in enumerate(SyntheticStream(50 + s, 30 + 10 * s, F, parts=K, dim=D, with_embeddings=True)):
            d = fr["dets"]
            n = len(d) if f % 7 != 3 else 0
            ids[s, f, :n] = d[:n, 6]
            ltwh[s, f, :n] = ltrb_to_ltwh_rows(d[:n, :4])
            emb[s, f, :n] = fr["embeddings"][:n]
            vis[s, f, :n] = fr["visibility"][:n]
            conf[s, f, :n] = d[:n, 4]
            counts[s, f] = n
            exp[s][f] = ref.update(ids[s, f, :n], ltwh[s, f, :n], emb[s, f, :n], vis[s, f, :n], conf[s, f, :n]) if n else None
    t = [torch.from_numpy(a).cuda() for a in (ids, ltwh, emb, vis, conf, counts)]
    rows = torch.zeros((S, F, MAXD, BPBSS_ROW.itemsize), dtype=torch.uint8, device="cuda")
    oc = torch.zeros((S, F), dtype=torch.int32, device="cuda")
    torch.cuda.synchronize()
    bank.update_dev(*[x.data_ptr() for x in t], F, rows.data_ptr(), MAXD, oc.data_ptr(), None)
    torch.cuda.synchronize()
    r = rows.cpu().numpy().view(BPBSS_ROW).reshape(S, F, MAXD)
    o = oc.cpu().numpy()
    for s in range(S):
        for f in range(F):
            e = exp[s][f]
            if e is None:
                assert o[s, f] == 0
                continue
            assert o[s, f] == len(e), (s, f, o[s, f], len(e))
            got = r[s, f, :len(e)]
            for name in ("det_id", "track_id", "hits", "age", "tsu", "state", "matched_name"):
                np.testing.assert_array_equal(got[name], e[name], err_msg=f"{s},{f} {name}")
    bank.close()


@pytest.mark.parametrize("T,N,K,D", [(100, 100, 6, 256), (37, 111, 6, 512), (1, 1, 1, 16), (130, 70, 8, 64)])
def test_partdist_mfma_matches_oracle(orc, T, N, K, D):
    import torch
    from tracklab_amd import _lib
    rng = np.random.default_rng(T * N + D)
    proto = rng.normal(0, 1, (max(T, N), K, D)).astype(np.float32)
    q = (proto[:T] + 0.1 * rng.normal(0, 1, (T, K, D))).astype(np.float32)
    g = (proto[:N] + 0.1 * rng.normal(0, 1, (N, K, D))).astype(np.float32)
    qv = (rng.uniform(0, 1, (T, K)) < 0.8).astype(np.uint8)
    gv = (rng.uniform(0, 1, (N, K)) < 0.8).astype(np.uint8)
    if T > 2:
        qv[1] = 0       # a track with no visible part: distance -0.5 everywhere (reference quirk kept)
    exp = orc.partdist(q, qv, g, gv)
    out = _lib.partdist(torch.from_numpy(q).cuda(), torch.from_numpy(qv).cuda(), torch.from_numpy(g).cuda(),
                        torch.from_numpy(gv).cuda())
    torch.cuda.synchronize()
    np.testing.assert_allclose(out.cpu().numpy(), exp, rtol=1e-5, atol=1e-5)   # fp32 contraction, cancellation-amplified


def test_hip_bpbss_beyond_512_tracks(orc):
    """max_tracks up to 1024 (the reference's track list grows; max_age 300 keeps every lost track for 10 s): four
    different 230-object scenes shown in turn leave ~650 live + coasting tracks, so the first-stage matrix has more rows than the
    register-resident Hungarian solver holds (512) and wave_lsa_lds takes over; then the first scene returns and is re-identified."""
    from tracklab_amd.synth import SyntheticStream, ltrb_to_ltwh_rows
    K, D = 6, 32
    bank = _bank(YAML, K, D, max_tracks=1024, max_dets=256)
    ref = orc.StrongSORT(K, D, **YAML)
    scenes = [iter(SyntheticStream(70 + k, 230, 8, parts=K, dim=D, with_embeddings=True, miss_prob=0.1)) for k in range(4)]
    most = 0
    for f, k in enumerate([0, 0, 1, 1, 2, 2, 3, 3, 0, 0, 1, 2, 3]):
        fr = next(scenes[k])
        d = fr["dets"]
        ltwh, conf, ids = ltrb_to_ltwh_rows(d[:, :4]), d[:, 4], (d[:, 6] + 100000 * k).astype(np.int64)
        exp = ref.update(ids, ltwh, fr["embeddings"], fr["visibility"], conf)
        got = bank.update(ids, ltwh, fr["embeddings"], fr["visibility"], conf)
        assert len(got) == len(exp), f
        for name in ("det_id", "track_id", "hits", "age", "tsu", "state", "matched_name", "pred_valid"):
            np.testing.assert_array_equal(got[name], exp[name], err_msg=f"{name} frame {f}")
        np.testing.assert_array_equal(got["kf_ltwh"], exp["kf_ltwh"])
        most = max(most, len(bank.tracks()[0]))
    assert most > 600
    with pytest.raises(Exception):
        _bank(YAML, K, D, max_tracks=16385)
    bank.close()


def test_hip_bpbss_2000_tracks_400_detections(orc):
    """Capacity is an allocation size (r04; the reference's lists grow: sort/tracker.py:427-441): twelve different 400-object
    scenes shown in turn leave > 2000 live + coasting tracks, 400 detections per frame -- beyond every LDS tier, so the per-frame lists and
    the Hungarian solver's work area are carved out of HBM -- and the rows still equal the oracle's, every frame; then a SMALL scene in the
    same bank (back in the LDS tier) and the return of the first scene (re-identification across 2000 tracks)."""
    from tracklab_amd.synth import SyntheticStream, ltrb_to_ltwh_rows
    K, D = 6, 32
    bank = _bank(YAML, K, D, max_tracks=4096, max_dets=512)
    ref = orc.StrongSORT(K, D, **YAML)
    scenes = [iter(SyntheticStream(170 + k, 400, 6, parts=K, dim=D, with_embeddings=True, miss_prob=0.05)) for k in range(12)]
    small = iter(SyntheticStream(990, 20, 4, parts=K, dim=D, with_embeddings=True))
    most = 0
    for f, k in enumerate([0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, -1, -1, 0]):
        fr = next(small) if k < 0 else next(scenes[k])
        d = fr["dets"]
        ltwh, conf, ids = ltrb_to_ltwh_rows(d[:, :4]), d[:, 4], (d[:, 6] + 100000 * (k + 1)).astype(np.int64)
        exp = ref.update(ids, ltwh, fr["embeddings"], fr["visibility"], conf)
        got = bank.update(ids, ltwh, fr["embeddings"], fr["visibility"], conf)
        assert len(got) == len(exp), f
        for name in ("det_id", "track_id", "hits", "age", "tsu", "state", "matched_name", "pred_valid"):
            np.testing.assert_array_equal(got[name], exp[name], err_msg=f"{name} frame {f}")
        np.testing.assert_array_equal(got["kf_ltwh"], exp["kf_ltwh"])
        most = max(most, len(bank.tracks()[0]))
    assert most > 2000
    bank.close()
