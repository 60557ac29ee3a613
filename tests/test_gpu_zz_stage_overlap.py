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
