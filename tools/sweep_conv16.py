"""Probe (not product): every distinct 16-bit convolution (f16 or split precision) of a network, timed under the tile configuration the heuristic of
tlk_conv2d_nhwc_16 picks and under every configuration tlk_conv16_set_config can force (-1: the r04 kernels, 1..26 f16 / 1..12 split).

    python tools/sweep_conv16.py f16 reid 2211        # part-based ReID ResNet-50, f16
    python tools/sweep_conv16.py split reid 2211      # the same in split precision (scaled (hi, lo) planes)
    python tools/sweep_conv16.py f16 yolox-m 24
"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tracklab_amd import _lib  # noqa: E402

mode = sys.argv[1] if len(sys.argv) > 1 else "f16"
what = sys.argv[2] if len(sys.argv) > 2 else "reid"
batch = int(sys.argv[3]) if len(sys.argv) > 3 else 2211
MIN_GAIN = float(os.environ.get("SWEEP_MIN_GAIN", "0.05"))
dev = torch.device("cuda:0")
split = mode == "split"

calls = []
real = _lib.conv2d_nhwc_16


def recorder(x, weight, bias=None, act=None, residual=None, stride=1, pad=None, x_lo=None, weight_lo=None, residual_lo=None, out_f32=False,
             residual_after_act=False, out=None, in_scale=None, res_scale=None, out_state=None):
    calls.append((tuple(x.shape), tuple(weight.shape), act, residual is not None, stride, pad, x_lo is not None, bool(out_f32), bool(residual_after_act),
                  in_scale is not None, res_scale is not None, out_state is not None))
    return real(x, weight, bias, act, residual, stride, pad, x_lo=x_lo, weight_lo=weight_lo, residual_lo=residual_lo, out_f32=out_f32,
                residual_after_act=residual_after_act, out=out, in_scale=in_scale, res_scale=res_scale, out_state=out_state)


_lib.conv2d_nhwc_16 = recorder
with torch.no_grad():
    if what.startswith("yolox"):
        from tracklab_amd.backbones.yolox import yolox
        net = yolox(what.split("-")[1], device=dev, dtype=torch.float32 if split else torch.float16)
        x = torch.rand(batch, 3, 640, 640, device=dev).contiguous(memory_format=torch.channels_last)
        net(x, split=True) if split else net(x.half())
    elif what.startswith("rtmpose"):
        from tracklab_amd.backbones.rtmpose import rtmpose
        net = rtmpose(what.split("-")[1], device=dev, dtype=torch.float16)
        x = torch.rand(batch, 3, 256, 192, device=dev).contiguous(memory_format=torch.channels_last)
        net(x.half())
    else:
        from tracklab_amd.backbones.reid import part_based_reid
        net = part_based_reid(6, 512, device=dev, dtype=torch.float32 if split else torch.float16, split_precision=split,
                              arch="hrnet32" if "hrnet" in what else "resnet50")
        x = torch.rand(batch, 3, 384, 128, device=dev).contiguous(memory_format=torch.channels_last)
        net(x if split else x.half())          # (split mode: the first forward calibrates the plane scales -- several passes)
        calls.clear()
        net(x if split else x.half())
torch.cuda.synchronize()
_lib.conv2d_nhwc_16 = real
shapes = {}
for c in calls:
    shapes.setdefault(c, [0])[0] += 1
print(f"{mode} {what} x {batch}: {sum(v[0] for v in shapes.values())} convolutions per forward, {len(shapes)} distinct shapes", flush=True)


def timed(fn, n):
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 0.02:
        fn()
        torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / n


L = _lib.lib()
_lib._bind_conv16(L)
CFGS = [-1] + list(range(1, 13 if split else 27))
tot_default = tot_best = 0.0
ONLY_COUT = int(os.environ.get("SWEEP_ONLY_COUT", "0"))      # e.g. 64: only the layers with that many output channels
for key, (count,) in sorted(shapes.items(), key=lambda kv: -kv[1][0]):
    if ONLY_COUT and key[1][0] != ONLY_COUT:
        continue
    xs, ws, act, res, stride, pad, sp, of32, raa, s_in, s_res, s_out = key
    cl = lambda t: t.contiguous(memory_format=torch.channels_last)          # noqa: E731
    x = cl(torch.randn(xs, device=dev).half()); xl = cl(torch.randn(xs, device=dev).half() * 0.01) if sp else None
    w = cl((torch.randn(ws, device=dev) * 0.05).half()); wl = cl((torch.randn(ws, device=dev) * 0.0005).half()) if sp else None
    b = torch.randn(ws[0], device=dev)
    one = lambda: torch.ones(1, device=dev)                                  # noqa: E731
    kw = dict(x_lo=xl, weight_lo=wl, out_f32=of32, residual_after_act=raa, in_scale=one() if s_in else None, res_scale=one() if (s_res and res) else None,
              out_state=torch.tensor([1.0, 0.0], device=dev) if s_out else None)
    y = real(x, w, b, act, None, stride, pad, **{**kw, "res_scale": None})
    r = rl = None
    if res:
        shp = y.shape if of32 else (y[0].shape if sp else y.shape)
        r = cl(torch.randn(shp, device=dev).half()); rl = cl(torch.randn(shp, device=dev).half() * 0.01) if sp else None
    out = None if of32 else y
    fn = lambda: real(x, w, b, act, r, stride, pad, residual_lo=rl, out=out, **kw)          # noqa: E731
    ho, wo = (y.shape if of32 or not sp else y[0].shape)[2:]
    flop = 2.0 * xs[0] * ho * wo * ws[0] * ws[1] * ws[2] * ws[3]
    n = max(3, min(50, int(20.0 / max(flop / (150e12 if sp else 400e12) * 1e3, 0.02))))
    L.tlk_conv16_set_config(0)
    timed(fn, n)
    t0 = timed(fn, n)
    best, row = (t0, "heuristic"), []
    for cfg in CFGS:
        if L.tlk_conv16_set_config(cfg) != 0:
            continue
        try:
            fn()
        except _lib.TlkError:
            continue
        t = timed(fn, n)
        row.append((cfg, t))
        if t < best[0]:
            best = (t, cfg)
    L.tlk_conv16_set_config(0)
    tot_default += count * t0
    tot_best += count * best[0]
    flag = "" if best[1] == "heuristic" or best[0] > t0 * (1 - MIN_GAIN) else f"  <-- cfg {best[1]}: {best[0]:.4f} ms ({(t0 / best[0] - 1) * 100:.0f} % faster)"
    print(f"{count:3d} {str(xs):>22s} {str(ws):>18s} s{stride} {str(act):>4s} res{int(res)} f32out{int(of32)} | {t0:8.4f} ms {flop / t0 / 1e9:6.1f} TF/s | "
          + " ".join(f"{c}:{t:.3f}" for c, t in sorted(row, key=lambda ct: ct[1])[:5]) + flag, flush=True)
print(f"sum over the network: heuristic {tot_default:.2f} ms, best forced per shape {tot_best:.2f} ms")
