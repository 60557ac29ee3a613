"""Where does a small step of the config3 pipeline spend its time?  For (n_streams, frames_per_step) in a few shapes: pipelined frames/s,
un-overlapped frame-in -> rows-out latency, and the GPU time of the stages of one step from HIP events on the main stream (detector graph,
decode + crops, ReID graph, hand-off copies) and on the association stream.  python tools/probe_latency.py [S,F ...]"""
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from tracklab_amd import gpu_pipeline as gp                              # noqa: E402
from tracklab_amd.synth import HEIGHT, WIDTH, SyntheticStream, render_frame, synth_yolox_head   # noqa: E402

shapes = [tuple(int(v) for v in a.split(",")) for a in sys.argv[1:]] or [(1, 1), (1, 2), (1, 4), (4, 1), (1, 8)]
dev = torch.device("cuda", 0)
out = []
for S, F in shapes:
    pipe = gp.DetReidTrackPipeline("m", n_streams=S, frames_per_step=F, max_dets=104)
    rng = np.random.default_rng(1)
    n_steps = 24
    streams = [list(SyntheticStream(100 + s, 100, n_steps * F)) for s in range(S)]
    heads = np.stack([np.stack([synth_yolox_head(rng, fr["dets"][:, :4], ratio=pipe.ratio) for fr in st]) for st in streams])     # (S, T, A, 6)
    heads = np.ascontiguousarray(heads.reshape(S, n_steps, F, -1, 6).transpose(1, 0, 2, 3, 4)).reshape(n_steps, S * F, -1, 6)
    d_heads = torch.from_numpy(heads).to(dev)
    frames = torch.from_numpy(np.stack([render_frame(rng, streams[s][f]["gt_boxes"]) for s in range(S) for f in range(F)])).to(dev)
    for k in range(8):
        pipe.step(frames, d_heads[k % n_steps])
    pipe.synchronize(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 60
    for k in range(n):
        pipe.step(frames, d_heads[k % n_steps])
    pipe.synchronize(); torch.cuda.synchronize()
    el = time.perf_counter() - t0
    lat = []
    for k in range(20):
        t1 = time.perf_counter(); pipe.step(frames, d_heads[k % n_steps]); pipe.synchronize(); lat.append(time.perf_counter() - t1)
    # host time of one step() call (launch overhead), GPU idle
    host = []
    for k in range(20):
        pipe.synchronize(); torch.cuda.synchronize()
        t1 = time.perf_counter(); pipe.step(frames, d_heads[k % n_steps]); host.append(time.perf_counter() - t1)
    pipe.synchronize()
    # GPU time of the main stream between step starts, un-overlapped
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    main_ms = []
    for k in range(10):
        pipe.synchronize(); torch.cuda.synchronize()
        e0.record(); pipe.step(frames, d_heads[k % n_steps]); e1.record(); torch.cuda.synchronize()
        main_ms.append(e0.elapsed_time(e1))
    row = {"n_streams": S, "frames_per_step": F, "fps": n * S * F / el, "ms_per_step_pipelined": el / n * 1e3,
           "ms_in_to_out": float(np.median(lat) * 1e3), "host_ms_per_step_call": float(np.median(host) * 1e3),
           "gpu_ms_main_stream_per_step": float(np.median(main_ms))}
    print(json.dumps(row), flush=True)
    out.append(row)
    pipe.close()
    del pipe
os.makedirs("gpurun_out", exist_ok=True)
json.dump(out, open("gpurun_out/probe_latency.json", "w"), indent=1)
