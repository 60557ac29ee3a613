"""Probe (not product): run a network's forward N times on the same input and report which module's output first differs between runs
(every kernel of libtlk is deterministic by construction: fixed summation order, no float atomics -- a difference means a race or a library
kernel that reduces with atomics).  python tools/probe_determinism.py rtmpose-t 300 [f16|f32]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

what = sys.argv[1] if len(sys.argv) > 1 else "rtmpose-t"
N = int(sys.argv[2]) if len(sys.argv) > 2 else 300
dt = torch.float32 if (len(sys.argv) > 3 and sys.argv[3] == "f32") else torch.float16
torch.manual_seed(0)
if what.startswith("rtmpose"):
    from tracklab_amd.backbones.rtmpose import rtmpose
    net = rtmpose(what.split("-")[1], device="cuda", dtype=dt)
    x = torch.randn(2, 3, 256, 192, device="cuda").to(dt).contiguous(memory_format=torch.channels_last)
elif what.startswith("yolox"):
    from tracklab_amd.backbones.yolox import yolox
    net = yolox(what.split("-")[1], device="cuda", dtype=dt)
    x = (torch.rand(2, 3, 640, 640, device="cuda") * 255).to(dt).contiguous(memory_format=torch.channels_last)
else:
    from tracklab_amd.backbones.reid import part_based_reid
    net = part_based_reid(6, 512, device="cuda", dtype=dt, arch="hrnet32" if "hrnet" in what else "resnet50")
    x = torch.randn(7, 3, 384, 128, device="cuda").to(dt).contiguous(memory_format=torch.channels_last)

outs = {}
order = []


def hook(name):
    def f(mod, inp, out):
        t = out[0] if isinstance(out, (tuple, list)) else out
        if torch.is_tensor(t):
            outs[name] = t.detach().clone()
            if name not in order:
                order.append(name)
    return f


for name, m in net.named_modules():
    if len(list(m.children())) == 0:
        m.register_forward_hook(hook(name))
with torch.no_grad():
    net(x)
    torch.cuda.synchronize()
    ref = {k: v.clone() for k, v in outs.items()}
    bad = {}
    for it in range(N):
        outs.clear()
        net(x)
        torch.cuda.synchronize()
        for k in order:
            if k in outs and not torch.equal(outs[k], ref[k]):
                bad.setdefault(k, []).append(it)
                break
print(f"{what} {dt}: {N} repeated forwards, {sum(len(v) for v in bad.values())} differed from the first")
for k, v in bad.items():
    print(f"  first differing module: {k} ({type(dict(net.named_modules())[k]).__name__}) in {len(v)} runs, e.g. runs {v[:5]}")
