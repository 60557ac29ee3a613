"""-m gpu: ByteTrack on the GPU (tlk_bytetrack_* through the C ABI) against the reference's golden runs and the oracle."""
import numpy as np
import pytest

from test_oracle_bytetrack import RUNS, replay

pytestmark = pytest.mark.gpu


class GpuTracker:
    def __init__(self, hp, **kw):
        from tracklab_amd._lib import ByteTrackBank
        self.bank = ByteTrackBank(**hp, **kw)

    def update(self, dets):
        r = self.bank.update(dets)
        return np.column_stack([r["ltrb"], r["track_id"], r["cls"], r["score"], r["det_id"]]).astype(np.float64).reshape(-1, 8)

    def tracks(self, which=0):
        return self.bank.tracks(which)


def check_lists(trk, g, f):
    for which, ln in ((0, "trk"), (1, "lost")):
        ids, mean, cov, st = trk.tracks(which)
        np.testing.assert_array_equal(ids, g[f"f{f}_{ln}_ids"])
        np.testing.assert_array_equal(st, g[f"f{f}_{ln}_state"])
        np.testing.assert_array_equal(mean, g[f"f{f}_{ln}_mean"])                  # library operation order (oracle/src/lapack_order.h): bit-exact
        np.testing.assert_array_equal(cov, g[f"f{f}_{ln}_cov"])


@pytest.mark.parametrize("name", RUNS)
def test_bytetrack_gpu_matches_reference(name):
    replay(name, lambda hp: GpuTracker(hp), check_lists)


def test_bytetrack_gpu_bit_exact_vs_oracle_and_min_confidence(orc):
    from tracklab_amd.synth import SyntheticStream
    hp = dict(track_thresh=0.55, match_thresh=0.85, track_buffer=8, frame_rate=30)
    gpu, cpu = GpuTracker(hp, min_confidence=0.3), orc.ByteTrack(**hp)
    for fr in SyntheticStream(12, 60, 120, miss_prob=0.1, churn_period=20, low_conf_frac=0.35):
        d = fr["dets"]
        a, b = gpu.update(d), cpu.update(d[d[:, 4] > 0.3])
        np.testing.assert_array_equal(a, b)                  # boxes bit-identical too: same fp64 / fp32 operation order
        for which in (0, 1):
            gi, gm, gc, gs = gpu.tracks(which)
            ci, cm, cc, cs = cpu.tracks(which)
            np.testing.assert_array_equal(gi, ci); np.testing.assert_array_equal(gs, cs)
            np.testing.assert_array_equal(gm, cm); np.testing.assert_array_equal(gc, cc)


def test_bytetrack_bank_batched_frames_and_reset():
    import torch
    from tracklab_amd._lib import BYTETRACK_ROW, ByteTrackBank
    from tracklab_amd.synth import SyntheticStream
    S, F, MAXD = 3, 15, 64
    hp = dict(track_thresh=0.5, track_buffer=5)
    bank = ByteTrackBank(**hp, n_streams=S, max_dets=MAXD, max_tracks=128, min_confidence=0.4, wrapper_mode=True)
    ref = [ByteTrackBank(**hp, max_dets=MAXD, max_tracks=128, min_confidence=0.4, wrapper_mode=True) for _ in range(S)]
    dets = np.zeros((S, F, MAXD, 7)); counts = np.zeros((S, F), np.int32)
    expect = [[None] * F for _ in range(S)]
    for s in range(S):
        for f, fr in enumerate(SyntheticStream(30 + s, 25, F, low_conf_frac=0.3, miss_prob=0.1)):
            d = fr["dets"] if (f + s) % 6 else fr["dets"][:0]          # some empty frames: wrapper_mode leaves the tracker untouched
            dets[s, f, :len(d)] = d; counts[s, f] = len(d)
            expect[s][f] = ref[s].update(d)
    cap = 128
    rows = torch.zeros((S, F, cap, BYTETRACK_ROW.itemsize), dtype=torch.uint8, device="cuda")
    ocnt = torch.zeros((S, F), dtype=torch.int32, device="cuda")
    bank.update_dev(torch.from_numpy(dets).cuda().data_ptr(), torch.from_numpy(counts).cuda().data_ptr(), F, rows.data_ptr(), cap, ocnt.data_ptr())
    torch.cuda.synchronize()
    got = rows.cpu().numpy().view(BYTETRACK_ROW).reshape(S, F, cap)
    oc = ocnt.cpu().numpy()
    for s in range(S):
        for f in range(F):
            e = expect[s][f]
            assert oc[s, f] == len(e)
            for name in BYTETRACK_ROW.names:
                np.testing.assert_array_equal(got[s, f, :len(e)][name], e[name])
    bank.reset(2)
    fr = next(iter(SyntheticStream(77, 6, 1)))
    r = bank.update(fr["dets"], stream=2)
    assert sorted(r["track_id"]) == list(range(1, len(r) + 1)) and len(r) > 0          # ids restart at 1; frame 1 tracks are activated at once


def test_bytetrack_rejects_bad_configuration():
    from tracklab_amd._lib import ByteTrackBank, TlkError
    with pytest.raises(TlkError):
        ByteTrackBank(max_tracks=20000)             # capacity is an allocation size up to 16384 tracks / 1024 detections per stream
    with pytest.raises(TlkError):
        ByteTrackBank(max_dets=2000)
    b = ByteTrackBank(max_dets=8)
    with pytest.raises(TlkError):
        b.update(np.zeros((9, 7)))


def test_bytetrack_400_tracks_300_detections(orc):
    """Capacity is an allocation size (r04; the reference's lists grow, byte_tracker.py:167-320): 300-object scenes shown in turn with a long
    track buffer leave almost 400 tracked + lost tracks beside 300 detections per frame -- past both LDS tiers, lists and the assignment problem
    in HBM -- rows, lists and Kalman state equal the oracle every frame; a small scene afterwards runs in the LDS tier again."""
    from tracklab_amd.synth import SyntheticStream
    hp = dict(track_thresh=0.5, match_thresh=0.8, track_buffer=60, frame_rate=30)
    gpu, cpu = GpuTracker(hp, max_tracks=4096, max_dets=512), orc.ByteTrack(**hp)
    scenes = [iter(SyntheticStream(300 + k, 300, 4, miss_prob=0.05, low_conf_frac=0.2)) for k in range(4)]
    small = iter(SyntheticStream(77, 20, 3))
    most = 0
    for f, k in enumerate([0, 0, 1, 1, 2, 2, 3, 3, 0, -1, -1, 1]):
        d = (next(small) if k < 0 else next(scenes[k]))["dets"]
        a, b = gpu.update(d), cpu.update(d)
        np.testing.assert_array_equal(a, b, err_msg=f"frame {f}")
        n = 0
        for which in (0, 1):
            gi, gm, gc, gs = gpu.tracks(which)
            ci, cm, cc, cs = cpu.tracks(which)
            np.testing.assert_array_equal(gi, ci); np.testing.assert_array_equal(gs, cs)
            np.testing.assert_array_equal(gm, cm); np.testing.assert_array_equal(gc, cc)
            n += len(gi)
        most = max(most, n)
    assert most > 350, most          # tracked + lost + 300 detections: beyond the 384 x 128 LDS tier
