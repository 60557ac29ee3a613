"""-m gpu: DetReidTrackPipeline(overlap_stages=True) against the serial pipeline.  In a file of its own that sorts LAST: the detector stage's
stream got its high priority (= a hardware queue of its own, real concurrency) in the last GPU minute of r05 -- after this test had passed with
both streams at normal priority -- so the driver's run is the first to execute it on truly concurrent queues; a failure here must not hide the
rest of the suite behind `-x`."""
import numpy as np
import pytest

from test_gpu_engine import _inputs

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("use_graph", [False, True])
def test_overlapped_stages_give_the_serial_pipeline_rows(use_graph):
    """DetReidTrackPipeline(overlap_stages=True), r05: detector stage of step t + 1 beside the ReID stage of step t on two streams, crops double
    buffered.  Steps are pushed back to back WITHOUT synchronising (that is where a missing dependency would show) and every step's rows must be
    the serial pipeline's, byte for byte."""
    import torch
    from tracklab_amd import gpu_pipeline as gp
    F, T = 1, 14
    rows = {}
    for overlap in (False, True):
        pipe = gp.DetReidTrackPipeline("s", n_streams=1, frames_per_step=F, max_dets=32, dim=64, use_graph=use_graph, overlap_stages=overlap)
        assert pipe.overlap == overlap
        heads, frames = _inputs(31, 14, T, pipe.ratio)
        d_heads = torch.from_numpy(heads).cuda()
        d_frames = [torch.from_numpy(np.stack(frames[t:t + F])).cuda() for t in range(0, T, F)]      # a buffer per step: nothing serialises the steps but the pipeline's own events
        got = []
        # (steps pushed back to back; the device is drained only at every third step, where that step's rows are read)
        for j in range(T // F):
            h_rows, h_cnt = pipe.step(d_frames[j], d_heads[j * F:(j + 1) * F])
            if j % 3 == 2:
                pipe.synchronize()
                got.append((h_rows.clone(), h_cnt.clone()))
        pipe.synchronize()
        got.append((h_rows.clone(), h_cnt.clone()))
        rows[overlap] = got
        pipe.close()
    assert len(rows[False]) == len(rows[True])
    for (r0, c0), (r1, c1) in zip(rows[False], rows[True]):
        assert torch.equal(c0, c1)
        assert torch.equal(r0, r1)


def test_auto_mode_measures_both_modes_and_is_never_slower_than_serial():
    """VERDICT r05 next 4 ("earn the default").  r06 finding: whether the two stage streams really run side by side is decided by HIP's stream ->
    hardware-queue -> command-processor-pipe assignment, which the pipeline cannot choose -- the same code ran 1.39x, 0.99x, 0.69x and 0.50x the
    serial pipeline in one process as other streams came and went (profiles/r06_overlap_autotune.md), and a spin-kernel probe of the two
    streams called all four "concurrent".  So the default is a MEASUREMENT: the first step of an eligible pipeline times both modes on its own
    inputs and keeps the faster.  Here, at the real online shape (YOLOX-m + ReID R50, 100 objects, one frame per step, f16, hipGraphs), with
    the stream assignment perturbed three ways: the trial ran, its decision follows its own numbers, the rows are those of a serial pipeline,
    and the steady-state rate is never below the serial pipeline's in the same process."""
    import time
    import torch
    from tracklab_amd import gpu_pipeline as gp
    heads, frames = _inputs(33, 100, 6, min(640 / 1080, 640 / 1920))
    d_heads = torch.from_numpy(heads).cuda()
    fr = torch.from_numpy(np.stack(frames[:1])).cuda()

    def run(pipe, n=100):
        rows = []
        for j in range(6):
            h_rows, h_cnt = pipe.step(fr, d_heads[j:j + 1])
            pipe.synchronize()
            rows.append(pipe.rows_numpy(h_rows, h_cnt)[0][0][0].copy())      # the VALID rows of the frame (the block behind them keeps whatever an earlier step left)
        for j in range(10):
            pipe.step(fr, d_heads[j % 6:j % 6 + 1], fetch=False)
        pipe.synchronize()
        t0 = time.perf_counter()
        for j in range(n):
            pipe.step(fr, d_heads[j % 6:j % 6 + 1], fetch=False)
        pipe.synchronize()
        return n / (time.perf_counter() - t0), rows
    keep, seen = [], []
    for extra in (0, 1, 4):
        keep += [torch.cuda.Stream(priority=(-1 if k % 2 else 0)) for k in range(extra)]          # shifts where the next streams land
        ps = gp.DetReidTrackPipeline("m", n_streams=1, frames_per_step=1, max_dets=104, dtype=torch.float16, overlap_stages=False)
        r_serial, rows_serial = run(ps)
        ps.close()
        pa = gp.DetReidTrackPipeline("m", n_streams=1, frames_per_step=1, max_dets=104, dtype=torch.float16)      # default = auto
        assert pa._ov_mode == "trial" and not pa.overlap
        r_auto, rows_auto = run(pa)
        tr = pa.overlap_trial
        assert tr is not None and pa._ov_mode in ("on", "off_tuned")
        rates3 = [tr["serial_steps_per_s"], tr["overlapped_steps_per_s"], tr["one_stream_steps_per_s"]]
        picked = 1 if pa.overlap else (2 if pa.trk_inline else 0)
        assert rates3[picked] >= max(rates3) / 1.06, (tr, picked)                    # the choice follows the trial's own numbers (5 % hysteresis towards the r05 arrangement)
        for r0, r1 in zip(rows_serial, rows_auto):                                               # the trial left no trace in the tracker: ids, boxes, states equal
            assert len(r0) == len(r1) > 50 and r0.tobytes() == r1.tobytes()
        seen.append((extra, round(r_serial, 1), round(r_auto, 1), pa.overlap, pa.overlap_note))
        pa.close()
        assert r_auto > 0.9 * r_serial, seen
    print("one frame per step, f16 (extra streams alive, serial frames/s, auto frames/s, overlapped?, note):")
    for row in seen:
        print("  ", row)


def test_auto_mode_is_for_the_online_16_bit_shapes_only():
    import torch
    from tracklab_amd import gpu_pipeline as gp
    p1 = gp.DetReidTrackPipeline("s", n_streams=1, frames_per_step=1, max_dets=16, dim=64, use_graph=False)
    p8 = gp.DetReidTrackPipeline("s", n_streams=1, frames_per_step=8, max_dets=16, dim=64, use_graph=False)
    p32 = gp.DetReidTrackPipeline("s", n_streams=1, frames_per_step=1, max_dets=16, dim=64, use_graph=False, dtype=torch.float32)
    assert p1._ov_mode == "trial" and p8._ov_mode == "off" and p32._ov_mode == "trial"        # (fp32 one-frame: the trial measures the overlap slower and drops it)
    assert not p8.overlap and p8.overlap_note == "off"
    p1.close(); p8.close(); p32.close()


def test_side_streams_are_picked_by_measured_concurrency():
    """r06: torch deals side streams from a pool and HIP deals each of them to one of a few hardware queues; a pipeline's association / copy / stage
    stream that lands on the compute stream's queue serialises behind it.  gpu_pipeline.pick_stream draws until the chain probe says the new
    stream runs beside the ones it must not collide with.  Negative control: a stream is never concurrent with itself."""
    import torch
    from tracklab_amd import gpu_pipeline as gp
    dev = torch.device("cuda", 0)
    cur = torch.cuda.current_stream(dev)
    s0 = torch.cuda.Stream(device=dev)
    assert not gp.streams_run_concurrently(s0, s0)
    keep = [torch.cuda.Stream(device=dev) for _ in range(3)]            # shift the pool position
    a = gp.pick_stream(dev, [cur])
    b = gp.pick_stream(dev, [cur, a])
    assert gp.streams_run_concurrently(a, cur) and gp.streams_run_concurrently(b, a) and gp.streams_run_concurrently(b, cur), gp._PICK_LOG[-4:]
    pipe = gp.DetReidTrackPipeline("s", n_streams=1, frames_per_step=8, max_dets=16, dim=64, use_graph=False)
    assert gp.streams_run_concurrently(pipe.trk_stream, cur)
    pipe.close()
    del keep
