"""Probe (not product): what the libraries reach on the fp32 convolutions of the config-3 step (ReID ResNet-50 @ 384x128 x 2400 crops, YOLOX-m @
640x640 x 24 frames): MIOpen (benchmark mode) channels-last / NCHW, hipBLASLt for the 1x1 ones, and the f16 route beside them.  Prints TFLOP/s
per shape and the flop-weighted time of the whole ReID forward, so the hand-written fp32 MFMA kernel has a number to beat."""
import os
import sys
import time

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tracklab_amd.backbones  # noqa: F401  (MIOpen env + benchmark mode)


def bench(fn, n=4):
    fn(); fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n


B = int(sys.argv[1]) if len(sys.argv) > 1 else 2400
# (name, cin, cout, k, stride, Hin, Win, count in one ResNet-50 forward)
R50 = [("conv1 7x7s2", 3, 64, 7, 2, 384, 128, 1),
       ("l1 1x1 64>64", 64, 64, 1, 1, 96, 32, 1), ("l1 1x1 256>64", 256, 64, 1, 1, 96, 32, 2), ("l1 3x3 64", 64, 64, 3, 1, 96, 32, 3),
       ("l1 1x1 64>256", 64, 256, 1, 1, 96, 32, 4),
       ("l2 1x1 256>128", 256, 128, 1, 1, 96, 32, 1), ("l2 3x3 128 s2", 128, 128, 3, 2, 96, 32, 1), ("l2 down 256>512 s2", 256, 512, 1, 2, 96, 32, 1),
       ("l2 1x1 512>128", 512, 128, 1, 1, 48, 16, 3), ("l2 3x3 128", 128, 128, 3, 1, 48, 16, 3), ("l2 1x1 128>512", 128, 512, 1, 1, 48, 16, 4),
       ("l3 1x1 512>256", 512, 256, 1, 1, 48, 16, 1), ("l3 3x3 256 s2", 256, 256, 3, 2, 48, 16, 1), ("l3 down 512>1024 s2", 512, 1024, 1, 2, 48, 16, 1),
       ("l3 1x1 1024>256", 1024, 256, 1, 1, 24, 8, 5), ("l3 3x3 256", 256, 256, 3, 1, 24, 8, 5), ("l3 1x1 256>1024", 256, 1024, 1, 1, 24, 8, 6),
       ("l4 1x1 1024>512", 1024, 512, 1, 1, 24, 8, 1), ("l4 down 1024>2048", 1024, 2048, 1, 1, 24, 8, 1),
       ("l4 1x1 2048>512", 2048, 512, 1, 1, 24, 8, 2), ("l4 3x3 512", 512, 512, 3, 1, 24, 8, 3), ("l4 1x1 512>2048", 512, 2048, 1, 1, 24, 8, 3),
       ("reduce 2048>256", 2048, 256, 1, 1, 24, 8, 1)]

tot = {}
print(f"B = {B} crops; columns: TFLOP/s  (ms per call)")
for name, cin, cout, k, s, H, W, cnt in R50:
    Ho, Wo = (H + 2 * (k // 2) - k) // s + 1, (W + 2 * (k // 2) - k) // s + 1
    flop = 2.0 * B * Ho * Wo * cout * cin * k * k
    row = {}
    for tag, dt, cl in (("f32 nhwc", torch.float32, True), ("f32 nchw", torch.float32, False), ("f16 nhwc", torch.float16, True)):
        x = torch.randn(B, cin, H, W, device="cuda", dtype=dt)
        w = torch.randn(cout, cin, k, k, device="cuda", dtype=dt) * 0.05
        if cl:
            x, w = x.contiguous(memory_format=torch.channels_last), w.contiguous(memory_format=torch.channels_last)
        try:
            row[tag] = bench(lambda: F.conv2d(x, w, None, s, k // 2))
        except Exception as e:      # noqa: BLE001
            row[tag] = float("nan"); print("   ", tag, str(e)[:100])
        if k == 1 and s == 1 and cl:
            x2, w2 = x.permute(0, 2, 3, 1).reshape(-1, cin), w.reshape(cout, cin)
            row[tag.split()[0] + " gemm"] = bench(lambda: F.linear(x2, w2))
        del x, w
    for t, v in row.items():
        tot[t] = tot.get(t, 0.0) + v * cnt
    if "f32 gemm" not in row:
        for t in ("f32 gemm", "f16 gemm"):
            tot[t] = tot.get(t, 0.0) + row[t.split()[0] + " nhwc"] * cnt
    print(f"{name:22s} x{cnt} {flop / 1e9:8.1f} GF: " + "  ".join(f"{t} {flop / v / 1e12:6.1f} ({v * 1e3:7.3f})" for t, v in row.items()), flush=True)
print("whole ResNet-50 forward, convolutions only, ms:", {t: round(v * 1e3, 2) for t, v in tot.items()})

# library ceilings: a large square fp32 / f16 GEMM
for dt in (torch.float32, torch.float16):
    a = torch.randn(8192, 8192, device="cuda", dtype=dt); b = torch.randn(8192, 8192, device="cuda", dtype=dt)
    t = bench(lambda: a @ b)
    print(f"GEMM 8192^3 {dt}: {2 * 8192 ** 3 / t / 1e12:.1f} TFLOP/s")
