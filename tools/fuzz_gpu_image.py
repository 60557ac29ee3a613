#!/usr/bin/env python3
"""Differential fuzz ON THE GPU BOX of the byte-moving kernels (letterbox, ReID crops with cv2 semantics, ReID crops with Pillow semantics, pose crops + SimCC decode) against
the C oracle on random frame sizes, box sets, output sizes, layouts and element types -- the r03 kernels (letterbox_wave_kernel, crop_wave3_kernel
with its 16-byte-pitch specialisation, pil_wave_kernel with its tap-count specialisations) have many shape-dependent paths. Pixels must be EQUAL
(integers 0..255 / the normalisation table are exact in f16 / bf16; fp32 bit for bit). The oracle is the checker here, never the product.

    python tools/fuzz_gpu_image.py [trials [family ...]]      # per kernel family: letterbox crop pil pose simcc
"""
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import oracle                                                             # noqa: E402
from tracklab_amd import _lib                                             # noqa: E402

oracle.build()
N = int(sys.argv[1]) if len(sys.argv) > 1 else 40
DT = {"f32": torch.float32, "f16": torch.float16, "bf16": torch.bfloat16}


def frames_of(rng, B, H, W):
    f = rng.integers(0, 256, (B, H, W, 3), dtype=np.uint8)
    f[:, ::7, ::5] = 255
    return f


def cast(exp32, dt):
    return torch.from_numpy(exp32).to(DT[dt]).float().numpy()


def fuzz_letterbox(t, rng):
    S = int(rng.choice([64, 320, 416, 640, 640, 640, 1280]))
    H, W = int(rng.integers(8, 1400)), int(rng.integers(8, 2200))
    if rng.random() < 0.4:
        W = int(rng.choice([640, 854, 960, 1280, 1920, 1918, 1284]))          # common widths, with and without a 16-byte row pitch
    B = int(rng.integers(1, 4))
    layout, dt = str(rng.choice(["focus_nhwc", "focus_nhwc", "nchw", "nhwc"])), str(rng.choice(["f16", "f16", "bf16", "f32"]))
    swap = bool(rng.random() < 0.5)
    fr = frames_of(rng, B, H, W)
    out, ratio = _lib.letterbox(torch.from_numpy(fr).cuda(), S, layout, DT[dt], swap_rb=swap)
    got = out.float().cpu().numpy()
    if layout == "focus_nhwc":
        g = np.empty((B, 3, S, S), dtype=np.float32)
        g[:, :, 0::2, 0::2] = got[:, 0:3]; g[:, :, 1::2, 0::2] = got[:, 3:6]; g[:, :, 0::2, 1::2] = got[:, 6:9]; g[:, :, 1::2, 1::2] = got[:, 9:12]
        got = g
    for b in range(B):
        exp, eratio = oracle.letterbox(fr[b][..., ::-1] if swap else fr[b], S)
        if ratio != eratio or not np.array_equal(got[b], exp):
            print(f"DIVERGENCE letterbox trial {t}: S {S} H {H} W {W} B {B} {layout} {dt} swap {swap} frame {b}: {int((got[b] != exp).sum())} values differ")
            return False
    return True


def fuzz_crop(t, rng):
    H, W = int(rng.integers(120, 1100)), int(rng.choice([640, 854, 960, 1280, 1920, 1918, 1284, int(rng.integers(200, 2000))]))
    B, MAXN = int(rng.integers(1, 3)), int(rng.integers(1, 24))
    oh, layout, dt = int(rng.choice([384, 256])), str(rng.choice(["nhwc", "nhwc", "nchw"])), str(rng.choice(["f16", "f16", "bf16", "f32"]))
    fr = frames_of(rng, B, H, W)
    boxes = np.zeros((B, MAXN, 4), dtype=np.float32)
    counts = rng.integers(0, MAXN + 1, B).astype(np.int32)
    for b in range(B):
        w, h = rng.uniform(1, 0.4 * W, MAXN), rng.uniform(1, 0.9 * H, MAXN)
        boxes[b] = np.stack([rng.uniform(-0.5 * w, W - 5), rng.uniform(-0.5 * h, H - 5), w, h], 1)       # (every box intersects the frame)
    if MAXN > 2:
        boxes[B - 1, 0] = [W - 70, H - 150, 200, 400]                           # the last rows of the last frame
        boxes[0, 1] = [rng.uniform(0, W - 3), rng.uniform(0, H - 3), 1.3, 1.2]   # tiny
    out = _lib.roi_crop_resize_norm(torch.from_numpy(fr).cuda(), torch.from_numpy(boxes).cuda(), torch.from_numpy(counts).cuda(), oh, 128, layout, DT[dt])
    got = out.float().cpu().numpy()
    for b in range(B):
        n = int(counts[b])
        ltrb = oracle.ltwh_to_crop_ltrb(boxes[b].astype(np.float64), W, H)
        exp = cast(oracle.crop_resize_norm(fr[b], ltrb, oh, 128)[:n], dt) if n else np.zeros((0, 3, oh, 128), np.float32)
        if not np.array_equal(got[b * MAXN:b * MAXN + n], exp) or got[b * MAXN + n:(b + 1) * MAXN].any():
            print(f"DIVERGENCE crop trial {t}: H {H} W {W} B {B} MAXN {MAXN} oh {oh} {layout} {dt} frame {b}")
            return False
    return True


def fuzz_pil(t, rng):
    H, W = int(rng.integers(120, 1100)), int(rng.choice([640, 854, 960, 1280, 1920, 1918, 1284, int(rng.integers(200, 2000))]))
    B, MAXN = int(rng.integers(1, 3)), int(rng.integers(1, 20))
    oh, dt = int(rng.choice([256, 256, 384, 130])), str(rng.choice(["f16", "f16", "bf16", "f32"]))
    layout = "nhwc" if dt != "f32" or rng.random() < 0.5 else "nchw"
    fr = frames_of(rng, B, H, W)
    xyxy = np.zeros((B, MAXN, 7))
    counts = rng.integers(0, MAXN + 1, B).astype(np.int32)
    for b in range(B):
        w, h = rng.uniform(1, 0.3 * W, MAXN), rng.uniform(1, 0.9 * H, MAXN)
        x, y = rng.uniform(-20, W - 5, MAXN), rng.uniform(-20, H - 5, MAXN)
        xyxy[b, :, :4] = np.stack([x, y, x + w, y + h], 1)
    if MAXN > 1:
        xyxy[B - 1, 0, :4] = [W - 120, H - 260, W + 10, H + 10]
    out = _lib.roi_crop_pil_resize_norm(torch.from_numpy(fr).cuda(), torch.from_numpy(xyxy).cuda(), torch.from_numpy(counts).cuda(), oh, 128, layout, DT[dt])
    got = out.float().cpu().numpy()
    for b in range(B):
        for i in range(int(counts[b])):
            exp, _ = oracle.ssort_reid_preprocess(fr[b], xyxy[b, i, :4], oh, 128)
            if not np.array_equal(got[b * MAXN + i], cast(exp, dt)):
                print(f"DIVERGENCE pil trial {t}: H {H} W {W} oh {oh} {layout} {dt} frame {b} box {i} {xyxy[b, i, :4]}")
                return False
        if got[b * MAXN + int(counts[b]):(b + 1) * MAXN].any():
            print(f"DIVERGENCE pil trial {t}: padding slots written"); return False
    return True


def fuzz_pose(t, rng):
    H, W = int(rng.integers(200, 1100)), int(rng.choice([640, 960, 1280, 1920, 1918, int(rng.integers(300, 2000))]))
    B, MAXN = int(rng.integers(1, 3)), int(rng.integers(1, 16))
    dt, layout = str(rng.choice(["f16", "f32"])), str(rng.choice(["nhwc", "nchw"]))
    if dt == "f16":
        layout = "nhwc"
    fr = frames_of(rng, B, H, W)
    xyxy = np.zeros((B, MAXN, 7))
    counts = rng.integers(0, MAXN + 1, B).astype(np.int32)
    for b in range(B):
        w, h = rng.uniform(4, 0.5 * W, MAXN), rng.uniform(4, 0.9 * H, MAXN)
        x, y = rng.uniform(-0.3 * w, W - 0.5 * w), rng.uniform(-0.3 * h, H - 0.5 * h)        # boxes that leave the image: constant-0 border
        xyxy[b, :, :4] = np.stack([x, y, x + w, y + h], 1)
    out, meta = _lib.pose_crop_warp_norm(torch.from_numpy(fr).cuda(), torch.from_numpy(xyxy).cuda(), torch.from_numpy(counts).cuda(), 192, 256, layout, DT[dt])
    got, meta = out.float().cpu().numpy(), meta.cpu().numpy()
    for b in range(B):
        for i in range(int(counts[b])):
            exp, c, sc = oracle.rtmpose_preprocess(fr[b], xyxy[b, i, :4])
            slot = b * MAXN + i
            if not (np.array_equal(meta[slot, :2], c) and np.array_equal(meta[slot, 2:4], sc) and np.array_equal(got[slot], cast(exp, dt))):
                print(f"DIVERGENCE pose trial {t}: H {H} W {W} {layout} {dt} frame {b} box {i} {xyxy[b, i, :4]}")
                return False
        if got[b * MAXN + int(counts[b]):(b + 1) * MAXN].any():
            print(f"DIVERGENCE pose trial {t}: padding slots written"); return False
    return True


def fuzz_simcc(t, rng):
    n, K, Wx, Wy = int(rng.integers(1, 60)), 17, 384, 512
    sx = rng.normal(0, 1, (n, K, Wx)).astype(np.float32); sy = rng.normal(0, 1, (n, K, Wy)).astype(np.float32)
    for _ in range(int(rng.integers(0, 4))):                                   # ties and non-positive maxima
        i, k = int(rng.integers(0, n)), int(rng.integers(0, K))
        if rng.random() < 0.5:
            sx[i, k] = -np.abs(sx[i, k])
        else:
            a, b2 = sorted(rng.integers(0, Wx, 2)); sx[i, k, a] = sx[i, k].max() + 1; sx[i, k, b2] = sx[i, k, a]
    meta = np.zeros((n, 10)); meta[:, 0] = rng.uniform(0, 1920, n); meta[:, 1] = rng.uniform(0, 1080, n)
    meta[:, 3] = rng.uniform(20, 900, n); meta[:, 2] = meta[:, 3] * 0.75
    out = _lib.simcc_decode(torch.from_numpy(sx).cuda(), torch.from_numpy(sy).cuda(), torch.from_numpy(meta).cuda())
    kps, scv = out["kps_xyc"].cpu().numpy(), out["scores"].cpu().numpy()
    for i in range(n):
        ek, es = oracle.simcc_decode(sx[i], sy[i], meta[i, :2], meta[i, 2:4])
        if not (np.array_equal(kps[i, :, :2], ek) and np.array_equal(scv[i], es)):
            print(f"DIVERGENCE simcc trial {t}: box {i}"); return False
    return True


out = {}
for name, fn in (("letterbox", fuzz_letterbox), ("crop", fuzz_crop), ("pil", fuzz_pil), ("pose", fuzz_pose), ("simcc", fuzz_simcc)):
    if len(sys.argv) > 2 and name not in sys.argv[2:]:
        continue
    t0 = time.time()
    ok = sum(bool(fn(t, np.random.default_rng(70000 + t))) for t in range(N))
    out[name] = {"trials": N, "identical": ok, "seconds": round(time.time() - t0, 1)}
    print(f"{name}: {ok}/{N} trials bit-identical to the oracle; {time.time() - t0:.0f} s", flush=True)
os.makedirs("gpurun_out", exist_ok=True)
json.dump(out, open("gpurun_out/fuzz_gpu_image.json", "w"), indent=1)
