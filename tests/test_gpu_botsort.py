"""-m gpu: BoT-SORT on the GPU (tlk_botsort_* through the C ABI) against the reference's golden runs and the oracle."""
import numpy as np
import pytest

from test_oracle_botsort import RUNS, replay

pytestmark = pytest.mark.gpu


class GpuTracker:
    def __init__(self, D, hp, **kw):
        from tracklab_amd._lib import BoTSORTBank
        self.bank = BoTSORTBank(D, **hp, **kw)

    def update(self, dets, feats):
        r = self.bank.update(dets, feats)
        return np.column_stack([r["ltrb"], r["track_id"], r["cls"], r["score"], r["det_id"]]).astype(np.float64).reshape(-1, 8)

    def tracks(self, which=0):
        return self.bank.tracks(which)


def check_lists(trk, g, f):
    for which, ln in ((0, "trk"), (1, "lost")):
        ids, mean, cov, st, feat = trk.tracks(which)
        np.testing.assert_array_equal(ids, g[f"f{f}_{ln}_ids"])
        np.testing.assert_array_equal(st, g[f"f{f}_{ln}_state"])
        np.testing.assert_array_equal(mean, g[f"f{f}_{ln}_mean"])                  # library operation order (oracle/src/lapack_order.h): bit-exact
        np.testing.assert_array_equal(cov, g[f"f{f}_{ln}_cov"])
        np.testing.assert_allclose(feat, g[f"f{f}_{ln}_feat"], rtol=0, atol=5e-7)


@pytest.mark.parametrize("name", RUNS)
def test_botsort_gpu_matches_reference(name):
    replay(name, lambda D, hp: GpuTracker(D, hp), check_lists)


@pytest.mark.parametrize("D", [32, 512])
def test_botsort_gpu_vs_oracle_and_min_confidence(orc, D):
    """Same stream through both; the KF state is bit-identical (same fp64 / fp32 operation order), the float32 features
    agree to the rounding of the norm's summation order (a wavefront reduction here, a serial loop in the oracle)."""
    from tracklab_amd.synth import SyntheticStream
    hp = dict(track_high_thresh=0.55, new_track_thresh=0.65, track_buffer=8, match_thresh=0.85, proximity_thresh=0.6, appearance_thresh=0.3,
              frame_rate=30, lambda_=0.97)
    gpu, cpu = GpuTracker(D, hp, min_confidence=0.3), orc.BoTSORT(D, **hp)
    rng = np.random.default_rng(5)
    for fr in SyntheticStream(12, 50, 90, parts=1, dim=D, with_embeddings=True, miss_prob=0.1, churn_period=20, low_conf_frac=0.35):
        d = fr["dets"].copy()
        d[:, 5] = rng.integers(0, 4, len(d))
        e = fr["embeddings"][:, 0, :].astype(np.float32)
        keep = d[:, 4] > 0.3
        a, b = gpu.update(d, e), cpu.update(d[keep], e[keep])
        np.testing.assert_array_equal(a, b)
        for which in (0, 1):
            gi, gm, gc, gs, gf = gpu.tracks(which)
            ci, cm, cc, cs, cf = cpu.tracks(which)
            np.testing.assert_array_equal(gi, ci); np.testing.assert_array_equal(gs, cs)
            np.testing.assert_array_equal(gm, cm); np.testing.assert_array_equal(gc, cc)
            np.testing.assert_allclose(gf, cf, rtol=0, atol=5e-7)


def test_botsort_bank_batched_frames_and_reset():
    import torch
    from tracklab_amd._lib import BOTSORT_ROW, BoTSORTBank
    from tracklab_amd.synth import SyntheticStream
    S, F, MAXD, D = 3, 15, 64, 64
    hp = dict(track_high_thresh=0.5, track_buffer=5)
    kw = dict(max_dets=MAXD, max_tracks=128, min_confidence=0.4, wrapper_mode=True)
    bank = BoTSORTBank(D, **hp, n_streams=S, **kw)
    ref = [BoTSORTBank(D, **hp, **kw) for _ in range(S)]
    dets = np.zeros((S, F, MAXD, 7)); feats = np.zeros((S, F, MAXD, D), np.float32); counts = np.zeros((S, F), np.int32)
    expect = [[None] * F for _ in range(S)]
    for s in range(S):
        for f, fr in enumerate(SyntheticStream(30 + s, 25, F, parts=1, dim=D, with_embeddings=True, low_conf_frac=0.3, miss_prob=0.1)):
            n = len(fr["dets"]) if (f + s) % 6 else 0               # some empty frames: wrapper_mode leaves the tracker untouched
            dets[s, f, :n] = fr["dets"][:n]; feats[s, f, :n] = fr["embeddings"][:n, 0, :]; counts[s, f] = n
            expect[s][f] = ref[s].update(dets[s, f, :n], feats[s, f, :n])
    cap = 128
    rows = torch.zeros((S, F, cap, BOTSORT_ROW.itemsize), dtype=torch.uint8, device="cuda")
    ocnt = torch.zeros((S, F), dtype=torch.int32, device="cuda")
    d_dets, d_feats, d_cnt = torch.from_numpy(dets).cuda(), torch.from_numpy(feats).cuda(), torch.from_numpy(counts).cuda()
    bank.update_dev(d_dets.data_ptr(), d_feats.data_ptr(), d_cnt.data_ptr(), F, rows.data_ptr(), cap, ocnt.data_ptr())
    torch.cuda.synchronize()
    got = rows.cpu().numpy().view(BOTSORT_ROW).reshape(S, F, cap)
    oc = ocnt.cpu().numpy()
    for s in range(S):
        for f in range(F):
            e = expect[s][f]
            assert oc[s, f] == len(e)
            for name in BOTSORT_ROW.names:
                np.testing.assert_array_equal(got[s, f, :len(e)][name], e[name])
    bank.reset(2)
    fr = next(iter(SyntheticStream(77, 6, 1, parts=1, dim=D, with_embeddings=True)))
    r = bank.update(fr["dets"], fr["embeddings"][:, 0, :], stream=2)
    assert sorted(r["track_id"]) == list(range(1, len(r) + 1)) and len(r) > 0


def test_botsort_rejects_bad_configuration():
    from tracklab_amd._lib import BoTSORTBank, TlkError
    with pytest.raises(TlkError):
        BoTSORTBank(64, max_tracks=20000)           # capacity is an allocation size up to 16384 tracks / 1024 detections per stream
    b = BoTSORTBank(64, cmc_method="sparseOptFlow")            # the reference's default: accepted, but every update must bring the frame's warp
    with pytest.raises(TlkError):
        b.update(np.zeros((1, 7)), np.zeros((1, 64), np.float32))
    with pytest.raises(ValueError):
        BoTSORTBank(64, cmc_method="nope")
    b = BoTSORTBank(32, max_dets=8)
    with pytest.raises(TlkError):
        b.update(np.zeros((9, 7)), np.zeros((9, 32), np.float32))


def test_botsort_600_tracks_300_detections(orc):
    """Capacity is an allocation size (r04; the reference's lists grow, bot_sort.py:279-440): 300-object scenes shown in turn with a long track
    buffer leave over 600 tracked + lost tracks and 300 detections per frame -- past both LDS tiers -- rows, lists, Kalman state and
    features equal the oracle every frame; a small scene afterwards runs in the LDS tier again."""
    from tracklab_amd.synth import SyntheticStream
    D = 32
    hp = dict(track_high_thresh=0.5, new_track_thresh=0.6, track_buffer=60, match_thresh=0.8, proximity_thresh=0.5, appearance_thresh=0.25,
              frame_rate=30, lambda_=0.98)
    gpu, cpu = GpuTracker(D, hp, max_tracks=4096, max_dets=512), orc.BoTSORT(D, **hp)
    scenes = [iter(SyntheticStream(300 + k, 300, 4, parts=1, dim=D, with_embeddings=True, miss_prob=0.05, low_conf_frac=0.2)) for k in range(4)]
    small = iter(SyntheticStream(77, 20, 3, parts=1, dim=D, with_embeddings=True))
    most = 0
    for f, k in enumerate([0, 0, 1, 1, 2, 2, 3, 3, 0, -1, -1, 1]):
        fr = next(small) if k < 0 else next(scenes[k])
        d, e = fr["dets"], fr["embeddings"][:, 0, :].astype(np.float32)
        a, b = gpu.update(d, e), cpu.update(d, e)
        np.testing.assert_array_equal(a, b, err_msg=f"frame {f}")
        n = 0
        for which in (0, 1):
            gi, gm, gc, gs, gf = gpu.tracks(which)
            ci, cm, cc, cs, cf = cpu.tracks(which)
            np.testing.assert_array_equal(gi, ci); np.testing.assert_array_equal(gs, cs)
            np.testing.assert_array_equal(gm, cm); np.testing.assert_array_equal(gc, cc)
            np.testing.assert_allclose(gf, cf, rtol=0, atol=5e-7)
            n += len(gi)
        most = max(most, n)
    assert most > 600, most
