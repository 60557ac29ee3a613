"""GPU box: random shapes through the convolution entry points of r05 / r06, against the oracle (fp32: bit for bit) / fp32 arithmetic on the
f16-rounded operands (16-bit kernels: one f16 ulp of slack).  Covers what the parametrized tests fix by hand: every tile configuration that
accepts the shape, channel slices (pixel strides), residual before / after the activation, the dynamic batch, ragged everything.
usage: python tools/fuzz_gpu_conv.py [seconds] [seed]"""
import os
import sys
import time
import numpy as np
import torch
import torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import oracle
from tracklab_amd import _lib

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
rng = np.random.default_rng(seed)
L = _lib.lib()
_lib.conv2d_nhwc_f32(torch.zeros(0, 4, 2, 2, device="cuda").contiguous(memory_format=torch.channels_last),
                     torch.zeros(4, 4, 1, 1, device="cuda").contiguous(memory_format=torch.channels_last))
stats = {"f32": 0, "f32_declined": 0, "f16": 0, "stem16": 0, "patch": 0, "patch64": 0, "split": 0, "dw": 0}
t0 = time.time()


def nhwc(a):
    return torch.from_numpy(a).cuda().permute(0, 3, 1, 2)


def fail(msg):
    print("DIVERGENCE:", msg, flush=True)
    sys.exit(1)


while time.time() - t0 < budget:
    kind = rng.choice(["f32", "f32", "patch", "patch64", "f16", "f16", "stem16", "split", "dw"])
    act = [None, "relu", "silu"][rng.integers(3)]
    if kind in ("f32", "patch", "patch64"):
        if kind == "patch64":                      # r06: the patch-resident kernel on 64 channels (two K steps per pixel), whole rows per 128-pixel tile
            wo = int(rng.choice([8, 16, 32, 64])); h = int(rng.choice([128 // wo, 256 // wo, 384 // wo])); n = int(rng.integers(1, 4))
            cin, cout, k, s = 64, int(rng.choice([64, 72, 128])), 3, 1
            w = wo
        elif kind == "patch":                        # shapes the patch-resident kernel takes: 3 x 3 / 1 on 32 channels, whole rows per tile
            wo = int(rng.choice([8, 16, 32, 64])); h = int(rng.choice([256 // wo, 512 // wo, 768 // wo])); n = int(rng.integers(1, 4))
            cin, cout, k, s = 32, int(rng.choice([32, 36, 64, 96])), 3, 1
            w = wo
        else:
            n, h, w = int(rng.integers(1, 4)), int(rng.integers(3, 20)), int(rng.integers(3, 20))
            cin = int(rng.choice([4, 8, 12, 32, 36, 64, 96, 128])); cout = int(rng.choice([4, 17, 32, 48, 64, 96, 130, 256]))
            k = int(rng.choice([1, 3, 5])); s = int(rng.choice([1, 2]))
        res = bool(rng.integers(2)); after = res and bool(rng.integers(2))
        x = rng.standard_normal((n, h, w, cin)).astype(np.float32)
        wt = (rng.standard_normal((cout, k, k, cin)) * 0.1).astype(np.float32)
        b = rng.standard_normal(cout).astype(np.float32)
        exp = oracle.conv2d_nhwc_f32(x, wt, b, None, stride=s, act=act)
        r = rng.standard_normal(exp.shape).astype(np.float32) if res else None
        if res:
            exp = oracle.conv2d_nhwc_f32(x, wt, b, r, stride=s, act=act, res_after_act=after)
        xt, wtt, rt = nhwc(x), nhwc(wt), (nhwc(r) if res else None)
        cfgs = [-1] + list(rng.choice([0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 21, 22, 23, 24, 25, 26, 27, 28, 29, 30, 31, 32, 33, 34, 35, 36, 37, 38, 39], size=4, replace=False))
        if kind == "patch64":
            cfgs = [-1, 0] + list(rng.choice([34, 35, 36, 37], size=3, replace=False))
        # a channel slice as the output (pixel stride > Cout) and a dynamic batch, sometimes
        slack = int(rng.choice([0, 0, 4, 16]))
        live = int(rng.integers(1, n + 1)) if rng.integers(3) == 0 else n
        for cfg in cfgs:
            if L.tlk_conv2d_set_config(int(cfg)) != 0:
                continue
            wide = torch.full((n, exp.shape[1], exp.shape[2], cout + slack), -7.0, device="cuda").permute(0, 3, 1, 2)
            out = wide[:, :cout]
            nl = torch.tensor([live], dtype=torch.int32, device="cuda")
            try:
                if live != n:
                    _lib.conv_set_dynamic_batch(nl)
                try:
                    _lib.conv2d_nhwc_f32(xt, wtt, torch.from_numpy(b).cuda(), act, rt, stride=s, out=out, residual_after_act=after)
                except _lib.TlkError:
                    stats["f32_declined"] += 1       # a forced configuration that does not take this shape says so
                    continue
            finally:
                _lib.conv_set_dynamic_batch(None)
                L.tlk_conv2d_set_config(-1)
            got = wide.permute(0, 2, 3, 1).cpu().numpy()
            tol_ok = np.array_equal(got[:live, ..., :cout], exp[:live]) if act != "silu" else np.allclose(got[:live, ..., :cout], exp[:live], rtol=2e-6, atol=2e-6)
            if not tol_ok:
                fail(f"fp32 cfg {cfg} last {L.tlk_conv2d_last_config()} case {(n, h, w, cin, cout, k, s, act, res, after, slack, live)} max diff {np.abs(got[:live, ..., :cout] - exp[:live]).max()}")
            if not (got[live:] == -7.0).all() or not (got[..., cout:] == -7.0).all():
                fail(f"fp32 cfg {cfg}: wrote outside its rows / channels, case {(n, h, w, cin, cout, k, s, slack, live)}")
            stats[kind] += 1
    elif kind == "f16":
        n, h, w = int(rng.integers(1, 4)), int(rng.integers(3, 24)), int(rng.integers(3, 24))
        cin = int(rng.choice([8, 32, 48, 64, 96, 128, 160, 192])); cout = int(rng.choice([8, 24, 64, 72, 128, 256, 264]))
        k = int(rng.choice([1, 3])); s = int(rng.choice([1, 2])); res = bool(rng.integers(2))
        x = torch.from_numpy(rng.standard_normal((n, h, w, cin)).astype(np.float32)).half().cuda().permute(0, 3, 1, 2)
        wt = torch.from_numpy((rng.standard_normal((cout, k, k, cin)) * 0.1).astype(np.float32)).half().cuda().permute(0, 3, 1, 2)
        b = torch.from_numpy(rng.standard_normal(cout).astype(np.float32)).cuda()
        ref = F.conv2d(x.float(), wt.float(), b, stride=s, padding=k // 2)
        r = torch.from_numpy(rng.standard_normal(tuple(ref.permute(0, 2, 3, 1).shape)).astype(np.float32)).half().cuda().permute(0, 3, 1, 2) if res else None
        if res:
            ref = ref + r.float()
        ref = torch.relu(ref) if act == "relu" else (F.silu(ref) if act == "silu" else ref)
        for cfg in [-1, 0] + list(rng.choice(np.arange(1, 27), size=4, replace=False)):      # (r06: 19..22 = the half-step tiles; a configuration that does not take the shape declines)
            if L.tlk_conv16_set_config(int(cfg)) != 0:
                continue
            try:
                try:
                    y = _lib.conv2d_nhwc_16(x, wt, b, act, r, s, k // 2)
                except _lib.TlkError:
                    continue
            finally:
                L.tlk_conv16_set_config(-1)
            err = (y.float() - ref).abs()
            if not bool((err <= 3e-3 * ref.abs() + 3e-3 * (x.float().abs().max() * wt.float().abs().max() * (cin * k * k) ** 0.5)).all()):
                fail(f"f16 cfg {cfg} case {(n, h, w, cin, cout, k, s, act, res)} max err {float(err.max())}")
            stats["f16"] += 1
    elif kind == "split":
        # r06: split mode with SCALED planes -- random magnitudes up to ~1e6 on the input, calibrated scales, against torch's fp64 convolution at the
        # exact-fp32 kernel's bound; every split tile configuration that takes the shape
        n, h, w = int(rng.integers(1, 4)), int(rng.integers(3, 20)), int(rng.integers(3, 20))
        cin = int(rng.choice([32, 64, 96, 128])); cout = int(rng.choice([64, 72, 128, 256])); k = int(rng.choice([1, 3])); s = int(rng.choice([1, 2]))
        res = bool(rng.integers(2)); mag = float(10.0 ** rng.uniform(0, 6))
        x = (torch.from_numpy(rng.standard_normal((n, h, w, cin)).astype(np.float32)) * mag).cuda().permute(0, 3, 1, 2)
        wt = torch.from_numpy((rng.standard_normal((cout, k, k, cin)) * 0.1).astype(np.float32)).cuda().permute(0, 3, 1, 2)
        b = torch.from_numpy(rng.standard_normal(cout).astype(np.float32) * mag).cuda()
        ref = F.conv2d(x.double(), wt.double(), b.double(), stride=s, padding=k // 2)
        bound = F.conv2d(x.double().abs(), wt.double().abs(), b.double().abs(), stride=s, padding=k // 2)
        r = (torch.from_numpy(rng.standard_normal(tuple(ref.permute(0, 2, 3, 1).shape)).astype(np.float32)) * mag).cuda().permute(0, 3, 1, 2) if res else None
        if res:
            ref, bound = ref + r.double(), bound + r.double().abs()
        ref = torch.relu(ref) if act == "relu" else ref
        states = torch.zeros((3, 2), device="cuda"); states[:, 0] = 1.0
        changed = torch.zeros(1, dtype=torch.int32, device="cuda")
        wh, wl = _lib.split_planes(wt)
        for cfg in [-1, 0] + list(rng.choice(np.arange(1, 13), size=3, replace=False)):
            if L.tlk_conv16_set_config(int(cfg)) != 0:
                continue
            try:
                y = None
                for _ in range(4):                     # calibration: run, update, until no scale grew
                    changed.zero_()
                    xh, xl = _lib.split_planes(x, state=states[0])
                    rh, rl = _lib.split_planes(r, state=states[1]) if res else (None, None)
                    try:
                        yy = _lib.conv2d_nhwc_16(xh, wh, b, "relu" if act == "relu" else None, rh, s, k // 2, x_lo=xl, weight_lo=wl, residual_lo=rl,
                                                 in_scale=states[0], res_scale=states[1] if res else None, out_state=states[2])
                    except _lib.TlkError:
                        yy = None
                        break
                    y = _lib.merge_planes(yy[0], yy[1], scale=states[2])
                    _lib.split_scale_update(states, changed)
                    if int(changed.item()) == 0:
                        break
            finally:
                L.tlk_conv16_set_config(-1)
            if y is None:
                continue
            err = (y.double() - ref).abs()
            if not bool(torch.isfinite(y).all()) or not bool((err <= 2e-6 * bound + 1e-30).all()):
                fail(f"split cfg {cfg} case {(n, h, w, cin, cout, k, s, act, res, mag)} max err / bound {float((err / bound).max())} scales {states[:, 0].tolist()}")
            stats["split"] += 1
    elif kind == "stem16":
        n, h, w = int(rng.integers(1, 4)), int(rng.integers(8, 90)), int(rng.integers(8, 150))
        k = int(rng.choice([7, 3])); cout = int(rng.choice([8, 32, 48, 64])); xp = int(rng.choice([3, 3, 4, 8]))
        pool = bool(rng.integers(2)) and (w + 2 * (k // 2) - k) // 2 + 1 <= 64
        xw = torch.from_numpy(rng.standard_normal((n, h, w, xp)).astype(np.float32)).half().cuda()
        x = xw[..., :3].permute(0, 3, 1, 2)
        wt = torch.from_numpy((rng.standard_normal((cout, k, k, 3)) * 0.1).astype(np.float32)).half().cuda().permute(0, 3, 1, 2)
        b = torch.from_numpy(rng.standard_normal(cout).astype(np.float32)).cuda()
        ref = F.conv2d(x.float(), wt.float(), b, stride=2, padding=k // 2)
        ref = torch.relu(ref) if act == "relu" else (F.silu(ref) if act == "silu" else ref)
        ref = ref.half().float()
        if pool:
            ref = F.max_pool2d(ref, 3, 2, 1)
        y = _lib.conv_stem16(x, _lib.conv_stem16_pack(wt), cout, k, b, act, pool=pool)
        err = (y.float() - ref).abs()
        if not bool((err <= 2e-3 * ref.abs() + 2e-3).all()):
            fail(f"stem16 case {(n, h, w, k, cout, xp, act, pool)} max err {float(err.max())}")
        stats["stem16"] += 1
    else:                                            # "dw", r06: depthwise k x k, every lane shape (columns per lane x rows ahead), channel slices, both element types
        n, h, w = int(rng.integers(1, 4)), int(rng.integers(1, 40)), int(rng.integers(1, 40))
        half = bool(rng.integers(2))
        c = int(rng.choice([8, 16, 24, 48, 96])) if half else int(rng.choice([4, 8, 12, 48, 100]))
        k = int(rng.choice([3, 5]))
        x = rng.standard_normal((n, h, w, c)).astype(np.float32)
        wt = (rng.standard_normal((k, k, c)) * 0.3).astype(np.float32)
        b = rng.standard_normal(c).astype(np.float32) if rng.integers(4) else None
        if half:
            x, wt = x.astype(np.float16), wt.astype(np.float16)
        exp = oracle.dwconv2d_nhwc_f32(x.astype(np.float32), wt.astype(np.float32), b, act)
        slack = int(rng.choice([0, 0, 8, 16]))
        for cfg in (0, 1, 2, 3, 4):
            _lib.dwconv_set_config(cfg)
            try:
                wide = torch.full((n, h, w, c + slack), -7.0, device="cuda", dtype=torch.float16 if half else torch.float32).permute(0, 3, 1, 2)
                _lib.dwconv2d_nhwc(nhwc(x), torch.from_numpy(wt).cuda(), torch.from_numpy(b).cuda() if b is not None else None, act, out=wide[:, :c])
            finally:
                _lib.dwconv_set_config(0)
            got = wide.permute(0, 2, 3, 1).float().cpu().numpy()
            if half:
                e16 = exp.astype(np.float16).astype(np.float32)
                ok = np.array_equal(got[..., :c], e16) if act != "silu" else bool((np.abs(got[..., :c] - exp) <= 0.5 * np.spacing(np.abs(e16).astype(np.float16)).astype(np.float32) + 4e-7 * np.abs(exp) + 1e-12).all())
            else:
                ok = np.array_equal(got[..., :c], exp) if act != "silu" else np.allclose(got[..., :c], exp, rtol=2e-6, atol=1e-6)
            if not ok:
                fail(f"dwconv cfg {cfg} case {(n, h, w, c, k, act, half, slack)} max diff {np.abs(got[..., :c] - exp).max()}")
            if not (got[..., c:] == -7.0).all():
                fail(f"dwconv cfg {cfg}: wrote outside its channels, case {(n, h, w, c, k, slack)}")
            stats["dw"] += 1
print(f"{sum(stats.values()) - stats['f32_declined']} launches compared in {time.time() - t0:.0f} s, 0 divergences: {stats}")
