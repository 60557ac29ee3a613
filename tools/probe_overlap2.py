"""GPU box: does gpu_pipeline.pick_concurrent_streams PREDICT whether the stage-overlap mode pays?  For several process states (extra streams
created before, an RCCL communicator alive) build the one-frame f16 pipeline serial and overlapped and print the picker's verdict beside the
measured frames/s.  usage: python tools/probe_overlap2.py [--dist] [extra streams ...]"""
import os
import sys
import time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tracklab_amd import gpu_pipeline as gp
from tracklab_amd.synth import SyntheticStream, render_frame, synth_yolox_head

args = [a for a in sys.argv[1:] if not a.startswith("--")]
if "--dist" in sys.argv:
    from tracklab_amd import dist as tdist
    d = tdist.init_single("nccl")
    t = torch.ones(1, device="cuda"); d.all_reduce(t); d.barrier()
    print("one-rank nccl group alive", flush=True)
rng = np.random.default_rng(0)
T = 6
ratio = min(640 / 1080, 640 / 1920)
heads, frames = [], []
for fr in SyntheticStream(0, 100, T):
    heads.append(synth_yolox_head(rng, fr["dets"][:, :4], ratio=ratio)); frames.append(render_frame(rng, fr["gt_boxes"]))
dh = torch.from_numpy(np.stack(heads)).cuda()
fr1 = torch.from_numpy(frames[0][None]).cuda()
keep = []


def rate(overlap):
    pipe = gp.DetReidTrackPipeline("m", n_streams=1, frames_per_step=1, max_dets=104, dtype=torch.float16, overlap_stages=overlap)
    for j in range(8):
        pipe.step(fr1, dh[j % T:j % T + 1], fetch=False)
    pipe.synchronize()
    t0 = time.perf_counter()
    for j in range(100):
        pipe.step(fr1, dh[j % T:j % T + 1], fetch=False)
    pipe.synchronize()
    r = 100 / (time.perf_counter() - t0)
    note = pipe.overlap_note
    pipe.close()
    return r, note


for extra in [int(a) for a in (args or ["0", "1", "2", "3", "5"])]:
    keep += [torch.cuda.Stream(priority=(-1 if k % 2 else 0)) for k in range(extra)]      # perturb the stream -> queue assignment
    for s_ in keep[-extra:] if extra else []:
        with torch.cuda.stream(s_):
            torch.zeros(1, device="cuda")
    rs, _ = rate(False)
    ro, note = rate(True)
    print(f"extra streams +{extra} (alive {len(keep)}): serial {rs:.1f}  overlap-asked {ro:.1f}  ratio {ro / rs:.2f}  picker: {note}  stream picks (draws, ok): {gp._PICK_LOG[-8:]}", flush=True)
