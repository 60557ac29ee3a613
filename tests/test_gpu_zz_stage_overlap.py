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
        assert pipe.overlap == overlap or (overlap and "no pair of streams" in pipe.overlap_note)      # (r06: only entered with streams observed to be concurrent)
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


def test_overlap_mode_is_entered_only_with_streams_observed_to_run_concurrently():
    """PROVE the concurrency the stage-overlap mode claims.  r06: the pipeline measures it itself before entering the mode
    (gpu_pipeline.pick_concurrent_streams: spin kernels on the two candidate streams, event timestamps on one clock) and stays serial when no
    pair of streams runs side by side -- the situation r05 could not tell apart from the real thing.  Here: (a) the measurement has a negative
    control (a stream is never concurrent with itself); (b) a pipeline that reports the mode as entered holds two streams that are concurrent
    when measured AGAIN; (c) at the real online shape (YOLOX-m + ReID R50, 100 objects, one frame per step, f16, hipGraphs) its steps are
    faster than the serial pipeline's in the same process -- the two stages do overlap."""
    import time
    import torch
    from tracklab_amd import gpu_pipeline as gp
    s0 = torch.cuda.Stream()
    assert not gp.streams_run_concurrently(s0, s0)
    heads, frames = _inputs(33, 100, 6, min(640 / 1080, 640 / 1920))
    d_heads = torch.from_numpy(heads).cuda()
    fr = torch.from_numpy(np.stack(frames[:1])).cuda()
    rate = {}
    for overlap in (False, True):
        pipe = gp.DetReidTrackPipeline("m", n_streams=1, frames_per_step=1, max_dets=104, dtype=torch.float16, overlap_stages=overlap)
        if overlap:
            if not pipe.overlap:
                assert "no pair of streams" in pipe.overlap_note
                pipe.close()
                pytest.skip("this device / process offers no two concurrent streams: the pipeline stayed serial (and said so) -- " + pipe.overlap_note)
            assert gp.streams_run_concurrently(pipe.det_stream, pipe.reid_stream), pipe.overlap_note
        for j in range(10):
            pipe.step(fr, d_heads[j % 6:j % 6 + 1], fetch=False)
        pipe.synchronize()
        t0 = time.perf_counter()
        for j in range(120):
            pipe.step(fr, d_heads[j % 6:j % 6 + 1], fetch=False)
        pipe.synchronize()
        rate[overlap] = 120 / (time.perf_counter() - t0)
        pipe.close()
    print(f"one frame per step, f16: serial {rate[False]:.1f} frames/s, stages overlapped {rate[True]:.1f} frames/s")
    assert rate[True] > 1.1 * rate[False], rate


def test_auto_mode_is_on_for_online_shapes_only():
    import torch
    from tracklab_amd import gpu_pipeline as gp
    p1 = gp.DetReidTrackPipeline("s", n_streams=1, frames_per_step=1, max_dets=16, dim=64, use_graph=False)
    p8 = gp.DetReidTrackPipeline("s", n_streams=1, frames_per_step=8, max_dets=16, dim=64, use_graph=False)
    assert (p1.overlap and p1.overlap_note.startswith("on")) or "no pair of streams" in p1.overlap_note
    assert not p8.overlap and p8.overlap_note == "off"
    p1.close(); p8.close()
