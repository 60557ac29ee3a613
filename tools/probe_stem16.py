"""GPU box: the f16 RGB stem of the ReID network (2400 crops of 384 x 128), fused kernel vs the two-pass library route it replaces.
usage: python tools/probe_stem16.py [crops]"""
import sys
import torch
import torch.nn.functional as F
sys.path.insert(0, ".")
from tracklab_amd import _lib

n = int(sys.argv[1]) if len(sys.argv) > 1 else 2400
x = torch.randn(n, 384, 128, 3, device="cuda").half().permute(0, 3, 1, 2)
w = (torch.randn(64, 7, 7, 3, device="cuda") * 0.1).half().permute(0, 3, 1, 2)
b = torch.randn(64, device="cuda")
packed = _lib.conv_stem16_pack(w)


def timeit(fn, it=10):
    fn(); fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / it


fused = lambda: _lib.conv_stem16(x, packed, 64, 7, b, "relu", pool=True)                 # noqa: E731
nopool = lambda: _lib.conv_stem16(x, packed, 64, 7, b, "relu", pool=False)               # noqa: E731
lib2 = lambda: _lib.maxpool2d_nhwc(torch.relu_(F.conv2d(x, w, b.half(), stride=2, padding=3)), 3, 2, 1)      # noqa: E731
gb_in, gb_out = n * 384 * 128 * 3 * 2 / 1e9, n * 96 * 32 * 64 * 2 / 1e9
t = timeit(fused)
print(f"{n} crops: stem + bias + ReLU + max-pool fused  {t:7.3f} ms  ({(gb_in + gb_out) / t * 1e3:6.0f} GB/s on {gb_in + gb_out:.2f} GB algorithmic, "
      f"{2 * n * 192 * 64 * 64 * 147 / t / 1e9:6.1f} TFLOP/s)")
t2 = timeit(nopool)
print(f"{n} crops: stem + bias + ReLU (no pool)         {t2:7.3f} ms  ({(gb_in + 4 * gb_out) / t2 * 1e3:6.0f} GB/s)")
t3 = timeit(lib2)
print(f"{n} crops: MIOpen conv + relu_ + tlk_maxpool    {t3:7.3f} ms")
y, r = fused(), lib2()
print("max |fused - library route|", float((y.float() - r.float()).abs().max()), "of max", float(r.float().abs().max()))
