"""-m gpu: the HIP OC-SORT path (libtlk.so through the C ABI) against (1) the golden vectors produced
by the reference and (2) the C oracle on fresh seeded streams, incl. the multi-stream / multi-frame
device entry point. ids / detection indices / row counts exact; fp64 boxes within 1e-9."""
import glob
import json
import os

import numpy as np
import pytest

from conftest import GOLDEN
from test_oracle_ocsort import run_ocsort

pytestmark = pytest.mark.gpu

OCSORT_FILES = sorted(glob.glob(os.path.join(GOLDEN, "ocsort_*.npz")))
YAML = {"min_confidence": 0.4, "hyper": dict(asso_func="giou", delta_t=1, det_thresh=0, inertia=0.3941737016672115,
                                             iou_threshold=0.22136877277096445, max_age=50, min_hits=1, use_byte=False)}


def _bank(cfg, **kw):
    from tracklab_amd._lib import OCSortBank
    return OCSortBank(**cfg["hyper"], min_confidence=cfg["min_confidence"], wrapper_mode=True, **kw)


@pytest.mark.parametrize("path", OCSORT_FILES, ids=[os.path.basename(p)[7:-4] for p in OCSORT_FILES])
def test_hip_ocsort_matches_reference_golden(path):
    g = np.load(path)
    cfg = json.loads(str(g["config"]))
    bank = _bank(cfg)

    def step(trk, dets, min_conf):
        return bank.update(dets, 0)

    def check_state(f, trk):
        if f"f{f}_kf_x" in g:
            x, P, ids = bank.tracks(0)
            np.testing.assert_array_equal(ids, g[f"f{f}_ids"])
            np.testing.assert_array_equal(x, g[f"f{f}_kf_x"])                     # library operation order, oracle/src/lapack_order.h
            np.testing.assert_array_equal(P, g[f"f{f}_kf_P"])

    run_ocsort(lambda **kw: None, step, g, check_state)
    bank.close()


@pytest.mark.parametrize("seed,nobj,kw", [(11, 100, {}), (12, 64, {"miss_prob": 0.2, "churn_period": 7}),
                                          (13, 5, {"miss_prob": 0.4})])
def test_hip_ocsort_matches_oracle_fresh_streams(orc, seed, nobj, kw):
    from tracklab_amd.synth import SyntheticStream
    cfg = YAML
    bank = _bank(cfg)
    ref = orc.OCSort(**cfg["hyper"])
    for fr in SyntheticStream(seed, nobj, 300, **kw):
        exp = orc.ocsort_wrapper_step(ref, fr["dets"], cfg["min_confidence"])
        got = bank.update(fr["dets"], 0)
        assert got.shape == exp.shape
        np.testing.assert_array_equal(got[:, [4, 5, 7]], exp[:, [4, 5, 7]])
        np.testing.assert_array_equal(got, exp)
    bank.close()


def test_hip_ocsort_reset_and_streams_are_independent(orc):
    """Two streams of one bank + reset(): same ids as two independent oracle trackers (reset restarts ids at 1,
    like OCSORT.reset() re-creating the tracker, oc_sort_api.py:28-30)."""
    from tracklab_amd.synth import SyntheticStream
    cfg = {"min_confidence": 0.4, "hyper": dict(asso_func="iou", delta_t=3, det_thresh=0.3, inertia=0.2,
                                                iou_threshold=0.3, max_age=30, min_hits=3, use_byte=False)}
    bank = _bank(cfg, n_streams=2)
    for rep in range(2):
        refs = [orc.OCSort(**cfg["hyper"]) for _ in range(2)]
        streams = [iter(SyntheticStream(21 + s + 10 * rep, 20, 60)) for s in range(2)]
        for f in range(60):
            for s in range(2):
                fr = next(streams[s])
                exp = orc.ocsort_wrapper_step(refs[s], fr["dets"], 0.4)
                got = bank.update(fr["dets"], s)
                assert got.shape == exp.shape
                np.testing.assert_array_equal(got[:, [4, 7]], exp[:, [4, 7]])
        bank.reset(-1)
    bank.close()


def test_hip_ocsort_device_batched_entry_point(orc):
    """tlk_ocsort_update_dev: S streams x F frames in one launch == per-frame oracle."""
    import torch
    from tracklab_amd.synth import SyntheticStream
    S, F, MAXD = 3, 50, 128
    cfg = YAML
    bank = _bank(cfg, n_streams=S, max_dets=MAXD)
    dets = np.zeros((S, F, MAXD, 7))
    counts = np.zeros((S, F), dtype=np.int32)
    exp = [[None] * F for _ in range(S)]
    for s in range(S):
        ref = orc.OCSort(**cfg["hyper"])
        for f, fr in enumerate(SyntheticStream(40 + s, 60 + 10 * s, F)):
            d = fr["dets"] if f % 17 != 3 else fr["dets"][:0]
            dets[s, f, :len(d)] = d
            counts[s, f] = len(d)
            exp[s][f] = orc.ocsort_wrapper_step(ref, d, 0.4)
    cap = 256
    d_dets = torch.from_numpy(dets).cuda()
    d_counts = torch.from_numpy(counts).cuda()
    d_out = torch.zeros((S, F, cap, 8), dtype=torch.float64, device="cuda")
    d_oc = torch.zeros((S, F), dtype=torch.int32, device="cuda")
    torch.cuda.synchronize()
    bank.update_dev(d_dets.data_ptr(), d_counts.data_ptr(), F, d_out.data_ptr(), cap, d_oc.data_ptr(), None)
    torch.cuda.synchronize()
    out, oc = d_out.cpu().numpy(), d_oc.cpu().numpy()
    for s in range(S):
        for f in range(F):
            e = exp[s][f]
            assert oc[s, f] == len(e), (s, f)
            np.testing.assert_array_equal(out[s, f, :len(e)][:, [4, 5, 7]], e[:, [4, 5, 7]])
            np.testing.assert_array_equal(out[s, f, :len(e)], e)
    bank.close()


def test_hip_ocsort_capacity_error_is_loud():
    from tracklab_amd._lib import OCSortBank, TlkError
    bank = OCSortBank(0.0, max_tracks=8, max_dets=16)
    dets = np.zeros((12, 7))
    dets[:, 0] = np.arange(12) * 100
    dets[:, 2] = dets[:, 0] + 50
    dets[:, 3] = 80
    dets[:, 4] = 0.9
    with pytest.raises(TlkError):
        bank.update(dets, 0)
    with pytest.raises(TlkError):
        bank.update(np.zeros((17, 7)), 0)
    bank.close()


def test_hip_ocsort_800_tracks_400_detections(orc):
    """Capacity is an allocation size (r04; the reference's list of trackers grows, oc_sort/ocsort.py:312-314): 400-object scenes shown in turn
    with max_age 60 leave over 800 live + coasting trackers and 400 detections per frame -- past both LDS tiers, lists and Hungarian work
    area in HBM -- rows and counts equal the oracle every frame; a small scene in the same bank afterwards runs in the LDS tier again."""
    from tracklab_amd._lib import OCSortBank
    from tracklab_amd.synth import SyntheticStream
    hyper = dict(asso_func="giou", delta_t=1, det_thresh=0.1, inertia=0.2, iou_threshold=0.3, max_age=60, min_hits=1, use_byte=True)
    bank = OCSortBank(**hyper, max_tracks=4096, max_dets=512)
    ref = orc.OCSort(**hyper)
    scenes = [iter(SyntheticStream(400 + k, 400, 4, miss_prob=0.05)) for k in range(6)]
    small = iter(SyntheticStream(77, 20, 3))
    most = 0
    for f, k in enumerate([0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, -1, -1, 0, 0, 1, 1, 5, 5]):
        d = (next(small) if k < 0 else next(scenes[k]))["dets"]
        exp = ref.update(d)
        got = bank.update(d, 0)
        assert got.shape == exp.shape, (f, got.shape, exp.shape)
        np.testing.assert_array_equal(got[:, [4, 5, 7]], exp[:, [4, 5, 7]], err_msg=f"ids frame {f}")
        np.testing.assert_allclose(got, exp, rtol=1e-11, atol=1e-9)
        gx, gP, gi = bank.tracks()
        cx, cP, ci = ref.tracks()
        np.testing.assert_array_equal(gi, ci, err_msg=f"track ids frame {f}")
        np.testing.assert_array_equal(gx, cx, err_msg=f"Kalman state frame {f}")      # also what update(None) / freeze left behind for EVERY unmatched tracker
        np.testing.assert_array_equal(gP, cP, err_msg=f"Kalman covariance frame {f}")
        most = max(most, len(gi))
    assert most > 700, most          # beyond the 512 x 256 LDS tier
    bank.close()
