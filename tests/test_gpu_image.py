"""-m gpu: letterbox / ROI crop-resize-normalize / YOLOX decode+NMS kernels vs the C oracle
(integer pixel values bit-exact; fp32 normalisation bit-exact; fp16/bf16 = rounded fp32)."""
import os

import numpy as np
import pytest

from conftest import GOLDEN

pytestmark = pytest.mark.gpu


def _frames(rng, B, H, W):
    f = rng.integers(0, 256, (B, H, W, 3), dtype=np.uint8)
    f[:, ::7, ::5] = 255
    return f


@pytest.mark.parametrize("H,W", [(1080, 1920), (720, 1280), (480, 854), (1000, 600), (333, 517)])
@pytest.mark.parametrize("layout", ["nchw", "nhwc", "focus_nhwc"])
def test_letterbox_matches_oracle(orc, H, W, layout):
    import torch
    from tracklab_amd import _lib
    rng = np.random.default_rng(H + W)
    frames = _frames(rng, 2, H, W)
    d = torch.from_numpy(frames).cuda()
    for dtype in (torch.float32, torch.float16, torch.bfloat16):
        out, ratio = _lib.letterbox(d, 640, layout, dtype)
        torch.cuda.synchronize()
        got = out.float().cpu().numpy()
        if layout == "focus_nhwc":       # undo YOLOX Focus: (tl, bl, tr, br) channel groups
            g = np.empty((2, 3, 640, 640), dtype=np.float32)
            g[:, :, 0::2, 0::2] = got[:, 0:3]
            g[:, :, 1::2, 0::2] = got[:, 3:6]
            g[:, :, 0::2, 1::2] = got[:, 6:9]
            g[:, :, 1::2, 1::2] = got[:, 9:12]
            got = g
        for b in range(2):
            exp, eratio = orc.letterbox(frames[b], 640)
            assert ratio == eratio
            np.testing.assert_array_equal(got[b], exp)       # integers 0..255 are exact in f16/bf16


@pytest.mark.parametrize("oh,ow", [(384, 128), (256, 128), (256, 192)])
@pytest.mark.parametrize("layout", ["nchw", "nhwc"])
def test_crop_resize_norm_matches_oracle(orc, oh, ow, layout):
    import torch
    from tracklab_amd import _lib
    rng = np.random.default_rng(oh * ow)
    B, H, W, MAXN = 2, 1080, 1920, 24
    frames = _frames(rng, B, H, W)
    boxes = np.zeros((B, MAXN, 4), dtype=np.float32)
    counts = np.array([MAXN, 17], dtype=np.int32)
    for b in range(B):
        x = rng.uniform(-30, W - 20, MAXN)
        y = rng.uniform(-30, H - 20, MAXN)
        w = rng.uniform(2, 300, MAXN)
        h = rng.uniform(2, 500, MAXN)
        boxes[b] = np.stack([x, y, w, h], 1)
    boxes[0, 0] = [0, 0, 5000, 5000]          # clipped to the full frame (downscale path)
    boxes[0, 1] = [100.5, 200.5, 1.2, 1.4]    # tiny box, half-to-even rounding
    boxes[0, 2] = [1918.7, 1078.2, 50, 50]    # at the border
    d = torch.from_numpy(frames).cuda()
    db = torch.from_numpy(boxes).cuda()
    dc = torch.from_numpy(counts).cuda()
    out32 = _lib.roi_crop_resize_norm(d, db, dc, oh, ow, layout, torch.float32)
    out16 = _lib.roi_crop_resize_norm(d, db, dc, oh, ow, layout, torch.float16)
    outbf = _lib.roi_crop_resize_norm(d, db, dc, oh, ow, layout, torch.bfloat16)
    torch.cuda.synchronize()
    got = out32.cpu().numpy()
    for b in range(B):
        ltrb = orc.ltwh_to_crop_ltrb(boxes[b].astype(np.float64), W, H)
        exp = orc.crop_resize_norm(frames[b], ltrb, oh, ow)
        n = counts[b]
        np.testing.assert_array_equal(got[b * MAXN:b * MAXN + n], exp[:n])
        assert not got[b * MAXN + n:(b + 1) * MAXN].any()
        e16 = torch.from_numpy(exp[:n]).half().numpy()
        np.testing.assert_array_equal(out16[b * MAXN:b * MAXN + n].cpu().numpy(), e16)
        ebf = torch.from_numpy(exp[:n]).bfloat16().float().numpy()
        np.testing.assert_array_equal(outbf[b * MAXN:b * MAXN + n].float().cpu().numpy(), ebf)


@pytest.mark.parametrize("oh,ow,dtype_name", [(384, 128, "float32"), (384, 128, "float16"), (256, 192, "float32")])
@pytest.mark.parametrize("layout", ["nchw", "nhwc"])
def test_dense_crop_batch_equals_the_slot_layout_row_for_row(oh, ow, dtype_name, layout):
    """r05: tlk_roi_crop_resize_norm_compact writes crop i of frame b at slot_base[b] + i (tlk_crop_slot_bases) -- the same pixels as the
    slot layout (itself bit-exact against the oracle above), no write outside the dense rows; every crop kernel (wave2 fp32, wave3 16-bit,
    the general one at 192 columns)."""
    import torch
    from tracklab_amd import _lib
    rng = np.random.default_rng(91)
    B, H, W, MAXN = 5, 1080, 1920, 24
    dtype = getattr(torch, dtype_name)
    frames = torch.from_numpy(_frames(rng, B, H, W)).cuda()
    boxes = np.stack([np.stack([rng.uniform(-30, W - 20, MAXN), rng.uniform(-30, H - 20, MAXN), rng.uniform(2, 300, MAXN), rng.uniform(2, 500, MAXN)], 1)
                      for _ in range(B)]).astype(np.float32)
    counts = np.array([MAXN, 0, 17, 1, 9], dtype=np.int32)
    db, dc = torch.from_numpy(boxes).cuda(), torch.from_numpy(counts).cuda()
    base, total, slot_of = _lib.crop_slot_bases(dc, MAXN)
    assert base.cpu().tolist() == [0, 24, 24, 41, 42] and int(total.item()) == 51
    so = slot_of.cpu().numpy().reshape(B, MAXN)
    for b in range(B):
        np.testing.assert_array_equal(so[b, :counts[b]], base[b].item() + np.arange(counts[b]))
        assert (so[b, counts[b]:] == 50).all()                    # padding slots point at a valid row
    slots = _lib.roi_crop_resize_norm(frames, db, dc, oh, ow, layout, dtype)
    dense = torch.full_like(slots, 7.0)
    _lib.roi_crop_resize_norm(frames, db, dc, oh, ow, layout, dtype, out=dense if layout == "nchw" else dense.permute(0, 2, 3, 1), slot_base=base)
    torch.cuda.synchronize()
    for b in range(B):
        n, b0 = int(counts[b]), int(base[b])
        assert torch.equal(dense[b0:b0 + n], slots[b * MAXN:b * MAXN + n])
    assert (dense[51:] == 7.0).all()


def test_crop_slot_bases_counts_beyond_one_wavefront():
    import torch
    from tracklab_amd import _lib
    rng = np.random.default_rng(3)
    counts = rng.integers(-2, 140, 200).astype(np.int32)          # negative / beyond max_n are clamped
    base, total, slot_of = _lib.crop_slot_bases(torch.from_numpy(counts).cuda(), 104)
    c = np.clip(counts, 0, 104)
    np.testing.assert_array_equal(base.cpu().numpy(), np.cumsum(c) - c)
    assert int(total.item()) == int(c.sum())
    so = slot_of.cpu().numpy().reshape(200, 104)
    for b in (0, 63, 64, 199):
        np.testing.assert_array_equal(so[b, :c[b]], (np.cumsum(c) - c)[b] + np.arange(c[b]))


from tracklab_amd.synth import synth_yolox_head as synth_head  # noqa: E402


@pytest.mark.parametrize("num_classes,nobj", [(1, 100), (1, 5), (3, 60), (1, 0)])
def test_yolox_decode_nms_matches_oracle(orc, num_classes, nobj):
    import torch
    from tracklab_amd import _lib
    from tracklab_amd.synth import SyntheticStream
    rng = np.random.default_rng(77 + nobj)
    B = 3
    preds = []
    for b in range(B):
        fr = SyntheticStream(100 + b, max(nobj, 1), 1).step()
        boxes = fr["dets"][:nobj, :4]
        preds.append(synth_head(rng, boxes, num_classes=num_classes))
    preds = np.stack(preds)
    ratio = np.float32(640 / 1920)
    d = torch.from_numpy(preds).cuda()
    out = _lib.yolox_decode_nms(d, 640, float(ratio), 1920, 1080, max_out=256)
    torch.cuda.synchronize()
    counts = out["counts"].cpu().numpy()
    for b in range(B):
        eb, es, ec = orc.yolox_postprocess(preds[b], 640, float(ratio))
        assert counts[b] == len(eb), (b, counts[b], len(eb))
        n = counts[b]
        if nobj >= 5:
            assert n >= nobj * 0.8
        np.testing.assert_array_equal(out["cls"][b, :n].cpu().numpy(), ec)
        np.testing.assert_allclose(out["scores"][b, :n].cpu().numpy(), es, rtol=0, atol=0)
        np.testing.assert_allclose(out["xyxy"][b, :n].cpu().numpy(), eb, rtol=2e-6, atol=1e-4)   # expf: 1-2 ulp
        # ltwh = sanitize_bbox_ltrb + ltrb_to_ltwh in float32 (coordinates.py:270-295,318-328)
        l = np.maximum(0, np.minimum(eb[:, 0], 1918)).astype(np.float32)
        t = np.maximum(0, np.minimum(eb[:, 1], 1078)).astype(np.float32)
        r = np.maximum(1, np.minimum(eb[:, 2], 1919)).astype(np.float32)
        bt = np.maximum(1, np.minimum(eb[:, 3], 1079)).astype(np.float32)
        exp_ltwh = np.stack([l, t, r - l, bt - t], 1)
        np.testing.assert_allclose(out["ltwh"][b, :n].cpu().numpy(), exp_ltwh, rtol=2e-6, atol=2e-4)


@pytest.mark.parametrize("frac,dup", [(0.04, 3), (0.10, 8), (0.20, 3), (0.40, 3)])
def test_yolox_decode_nms_with_many_candidates(orc, frac, dup):
    """r04 NMS (rank sort up to 1024 candidates, bitonic beyond; 64-row suppression bitmask blocks): hundreds to thousands of candidates per frame,
    heavy overlap -- a fraction of ALL anchors is lifted above the score threshold on top of 100 objects with `dup` near-duplicates each; kept
    set and order equal the oracle's sequential greedy loop (338, 850, 1700 and 3400 candidates: both sort paths, up to 54 blocks)"""
    import torch
    from tracklab_amd import _lib
    from tracklab_amd.synth import SyntheticStream
    rng = np.random.default_rng(int(frac * 100) + dup)
    preds = []
    for b in range(2):
        fr = SyntheticStream(300 + b, 100, 1).step()
        h = synth_head(rng, fr["dets"][:, :4], dup=dup)
        lift = rng.random(len(h)) < frac
        h[lift, 4] = rng.uniform(0.85, 1.0, int(lift.sum())).astype(np.float32)
        h[lift, 5] = rng.uniform(0.85, 1.0, int(lift.sum())).astype(np.float32)
        h[lift, 2:4] = rng.normal(1.2, 0.5, (int(lift.sum()), 2)).astype(np.float32)      # larger boxes: they overlap
        preds.append(h)
    preds = np.stack(preds)
    ratio = np.float32(640 / 1920)
    out = _lib.yolox_decode_nms(torch.from_numpy(preds).cuda(), 640, float(ratio), 1920, 1080, max_out=4096)
    counts = out["counts"].cpu().numpy()
    for b in range(2):
        eb, es, ec = orc.yolox_postprocess(preds[b], 640, float(ratio))
        ncand = int(((preds[b][:, 4] * preds[b][:, 5]) > 0.7).sum())
        assert ncand > (200 if frac < 0.1 else 1024 if frac >= 0.2 else 600) and len(eb) < ncand
        assert counts[b] == len(eb), (b, counts[b], len(eb), ncand)
        n = counts[b]
        np.testing.assert_array_equal(out["scores"][b, :n].cpu().numpy(), es)
        np.testing.assert_allclose(out["xyxy"][b, :n].cpu().numpy(), eb, rtol=2e-6, atol=1e-4)


@pytest.mark.parametrize("dtype_name", ["float16", "bfloat16"])
@pytest.mark.parametrize("act", ["relu", "silu", None])
def test_fused_bias_act_epilogue_matches_torch(dtype_name, act):
    import torch
    from tracklab_amd import _lib
    dtype = getattr(torch, dtype_name)
    g = torch.Generator(device="cuda").manual_seed(3)
    x = torch.randn((5, 72, 13, 11), generator=g, device="cuda").to(dtype).contiguous(memory_format=torch.channels_last)
    res = torch.randn((5, 72, 13, 11), generator=g, device="cuda").to(dtype).contiguous(memory_format=torch.channels_last)
    bias = torch.randn((72,), generator=g, device="cuda").to(dtype)
    for r in (None, res):
        ref = x.float() + bias.float().view(1, -1, 1, 1) + (r.float() if r is not None else 0)
        ref = torch.relu(ref) if act == "relu" else (torch.nn.functional.silu(ref) if act == "silu" else ref)
        got = _lib.bias_act_(x.clone(memory_format=torch.channels_last), bias, act, r)
        torch.cuda.synchronize()
        tol = 2e-3 if dtype == torch.float16 else 1.6e-2          # one rounding to the storage dtype (+ fast exp in SiLU)
        torch.testing.assert_close(got.float(), ref.to(dtype).float(), rtol=tol, atol=tol)


# ------------------------------------------------------------------------------------------------ plain StrongSORT ReID input (G1)
def test_pil_crop_resize_norm_golden_bit_exact():
    """Pillow-made golden: float32 output of every crop equals torch's (value, channel) normalisation table looked up at the
    resized uint8 Pillow produced -> the resize is bit-exact and so is the normalisation."""
    import torch
    from tracklab_amd import _lib
    g = np.load(os.path.join(GOLDEN, "pil_preprocess.npz"))
    img = torch.from_numpy(g["image"]).cuda()[None].contiguous()
    n = len(g["boxes"])
    boxes = torch.zeros((1, n + 2, 7), dtype=torch.float64, device="cuda")          # (n,7) detection rows: stride 7
    boxes[0, :n, :4] = torch.from_numpy(g["boxes"]).cuda()
    counts = torch.tensor([n], dtype=torch.int32, device="cuda")
    for layout in ("nchw", "nhwc"):
        out = _lib.roi_crop_pil_resize_norm(img, boxes, counts, 256, 128, layout, torch.float32).cpu().numpy()
        for i in range(n):
            exp = np.stack([g["norm_lut"][c][g["resized"][i][:, :, c]] for c in range(3)])
            np.testing.assert_array_equal(out[i], exp, err_msg=f"{layout} box {i}")
            if f"norm{i}" in g.files:
                np.testing.assert_array_equal(out[i], g[f"norm{i}"])
        assert not out[n:].any()                                                       # slots >= count are zero
    half = _lib.roi_crop_pil_resize_norm(img, boxes, counts, 256, 128, "nhwc", torch.float16).cpu().numpy()
    np.testing.assert_array_equal(half[0], g["norm0"].astype(np.float16))


@pytest.mark.parametrize("oh,ow", [(256, 128), (128, 64), (384, 128)])
def test_pil_crop_resize_norm_vs_oracle_1080p(orc, oh, ow):
    import torch
    from tracklab_amd import _lib
    from tracklab_amd.synth import SyntheticStream, render_frame
    rng = np.random.default_rng(3)
    fr = SyntheticStream(4, 60, 1).step()
    frame = render_frame(rng, fr["gt_boxes"])
    dets = fr["dets"].copy()
    dets[0, :4] = [5.2, 3.1, 700.7, 1000.9]            # far larger than the target: direct (unstaged) branch
    dets[1, :4] = [1900.0, 1000.0, 1950.0, 1100.0]     # clipped at the border
    dets[2, :4] = [300.0, 300.0, 300.4, 300.9]         # empty crop -> zeros
    n = len(dets)
    boxes = torch.from_numpy(dets[None]).cuda().contiguous()
    counts = torch.tensor([n], dtype=torch.int32, device="cuda")
    out = _lib.roi_crop_pil_resize_norm(torch.from_numpy(frame).cuda()[None].contiguous(), boxes, counts, oh, ow, "nchw", torch.float32).cpu().numpy()
    for i in range(n):
        exp, _ = orc.ssort_reid_preprocess(frame, dets[i, :4], oh, ow)
        np.testing.assert_array_equal(out[i], exp, err_msg=f"crop {i} box {dets[i, :4]}")


@pytest.mark.parametrize("layout", ["nchw", "nhwc", "focus_nhwc"])
def test_swap_rb_reads_the_frame_as_bgr(layout):
    """TLK_SWAP_RB (output channel c = source channel 2 - c, statistics indexed by OUTPUT channel) == the same kernel on the flipped frame:
    letterbox, both crop kernels and the pose warp. That is how the fused pipeline keeps ONE RGB frame in HBM for the ReID crops while the
    detector / pose estimator see BGR like the reference's cv2.imread (rtmlib_api.py:30)."""
    import torch
    from tracklab_amd import _lib
    rng = np.random.default_rng(77)
    B, H, W, MAXN = 2, 720, 1280, 12
    frames = _frames(rng, B, H, W)
    d, dflip = torch.from_numpy(frames).cuda(), torch.from_numpy(np.ascontiguousarray(frames[..., ::-1])).cuda()
    for dtype in (torch.float32, torch.float16):
        a, _ = _lib.letterbox(d, 640, layout, dtype, swap_rb=True)
        b, _ = _lib.letterbox(dflip, 640, layout, dtype)
        assert torch.equal(a, b)
    if layout == "focus_nhwc":
        return
    boxes = np.zeros((B, MAXN, 4), dtype=np.float32)
    boxes[..., 0] = rng.uniform(0, W - 200, (B, MAXN)); boxes[..., 1] = rng.uniform(0, H - 300, (B, MAXN))
    boxes[..., 2] = rng.uniform(20, 180, (B, MAXN)); boxes[..., 3] = rng.uniform(40, 290, (B, MAXN))
    boxes[0, 0] = [0, 0, 5000, 5000]                                   # direct (unstaged) branch
    counts = torch.tensor([MAXN, 7], dtype=torch.int32).cuda()
    db = torch.from_numpy(boxes).cuda()
    xyxy = torch.from_numpy(np.concatenate([boxes[..., :2], boxes[..., :2] + boxes[..., 2:]], axis=-1).astype(np.float64)).cuda().contiguous()
    for dtype in (torch.float32, torch.float16):
        assert torch.equal(_lib.roi_crop_resize_norm(d, db, counts, 384, 128, layout, dtype, swap_rb=True),
                           _lib.roi_crop_resize_norm(dflip, db, counts, 384, 128, layout, dtype))
        assert torch.equal(_lib.roi_crop_pil_resize_norm(d, xyxy, counts, 256, 128, layout, dtype, swap_rb=True),
                           _lib.roi_crop_pil_resize_norm(dflip, xyxy, counts, 256, 128, layout, dtype))
        pa, ma = _lib.pose_crop_warp_norm(d, xyxy, counts, 192, 256, layout, dtype, swap_rb=True)
        pb, mb = _lib.pose_crop_warp_norm(dflip, xyxy, counts, 192, 256, layout, dtype)
        assert torch.equal(pa, pb) and torch.equal(ma, mb)


@pytest.mark.parametrize("W", [1918, 1284, 854])
def test_crop_resize_norm_on_row_pitches_that_are_not_multiples_of_16_bytes(orc, W):
    """crop_wave3_kernel has a specialisation for frames whose row pitch W * 3 is a multiple of 16 bytes (every 1080p / 720p test above); these
    widths take its general-pitch code: per-row misalignment of the staged rows, per-row tap window shifts."""
    import torch
    from tracklab_amd import _lib
    rng = np.random.default_rng(W)
    B, H, MAXN = 2, 720, 20
    frames = _frames(rng, B, H, W)
    boxes = np.zeros((B, MAXN, 4), dtype=np.float32)
    counts = np.array([MAXN, 13], dtype=np.int32)
    for b in range(B):
        boxes[b] = np.stack([rng.uniform(-20, W - 20, MAXN), rng.uniform(-20, H - 20, MAXN), rng.uniform(2, 150, MAXN), rng.uniform(2, 400, MAXN)], 1)
    boxes[1, 0] = [W - 90, H - 200, 150, 300]           # the last rows of the last frame
    d, db, dc = torch.from_numpy(frames).cuda(), torch.from_numpy(boxes).cuda(), torch.from_numpy(counts).cuda()
    for swap in (False, True):
        out16 = _lib.roi_crop_resize_norm(d, db, dc, 384, 128, "nhwc", torch.float16, swap_rb=swap)
        out32 = _lib.roi_crop_resize_norm(d, db, dc, 384, 128, "nhwc", torch.float32, swap_rb=swap)
        torch.cuda.synchronize()
        assert torch.equal(out16, out32.half())
    got = _lib.roi_crop_resize_norm(d, db, dc, 384, 128, "nhwc", torch.float16).cpu().numpy()
    for b in range(B):
        ltrb = orc.ltwh_to_crop_ltrb(boxes[b].astype(np.float64), W, H)
        exp = orc.crop_resize_norm(frames[b], ltrb, 384, 128)
        n = counts[b]
        np.testing.assert_array_equal(got[b * MAXN:b * MAXN + n], torch.from_numpy(exp[:n]).half().numpy())


def test_the_general_pitch_code_of_the_crop_kernel_stays_bit_exact():
    """crop_wave3_kernel (128-wide 16-bit targets) takes a specialised path when the frame's row pitch is a multiple of 16 bytes (every 1080p
    test frame); its general-pitch code stays selectable with TLK_CROP_P16=0 (read once per process, hence the subprocess) and must produce the
    oracle's bits too.  (r04: the five superseded kernel generations and their switches are gone; shapes other than 128 columns run crop_kernel.)"""
    import subprocess
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(here, "test_gpu_image.py"), "-m", "gpu", "-q", "-x", "-k", "test_crop_resize_norm_matches_oracle and 128"],
                       env=dict(os.environ, TLK_CROP_P16="0"), capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and " passed" in r.stdout and "failed" not in r.stdout, r.stdout[-2000:] + r.stderr[-1000:]


def test_the_older_letterbox_kernel_stays_bit_exact():
    """letterbox_wave_kernel (r03) is the default for the detector's input format (Focus layout, 16-bit elements); letterbox_lds_kernel stays
    selectable with TLK_LETTERBOX_WAVE=0 (read once per process, hence the subprocess) and must keep producing the oracle's bits."""
    import subprocess
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(here, "test_gpu_image.py"), "-m", "gpu", "-q", "-x", "-k", "test_letterbox_matches_oracle and focus"],
                       env=dict(os.environ, TLK_LETTERBOX_WAVE="0"), capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and " passed" in r.stdout and "failed" not in r.stdout, r.stdout[-2000:] + r.stderr[-1000:]


@pytest.mark.parametrize("H,W,S", [(1080, 1920, 640), (1080, 1918, 640), (719, 1279, 640), (2160, 3840, 640), (540, 960, 1280), (97, 2000, 416), (2000, 97, 416), (6, 8, 64)])
def test_letterbox_focus_f16_odd_shapes(orc, H, W, S):
    """Shapes that stress letterbox_wave_kernel's edges: source rows that are not multiples of 16 bytes, an odd number of real rows (the bottom
    row of the last real pair is padding), a real width that ends inside / before a 256-column range, frames too wide for its staging rows
    (4K: falls back to letterbox_lds_kernel), up-scaling, tiny frames; 3 frames so that the last frame's last rows end at the end of the buffer."""
    import torch
    from tracklab_amd import _lib
    rng = np.random.default_rng(H * 7 + W)
    frames = _frames(rng, 3, H, W)
    d = torch.from_numpy(frames).cuda()
    for dtype in (torch.float16, torch.bfloat16):
        out, ratio = _lib.letterbox(d, S, "focus_nhwc", dtype)
        torch.cuda.synchronize()
        got = out.float().cpu().numpy()
        g = np.empty((3, 3, S, S), dtype=np.float32)
        g[:, :, 0::2, 0::2] = got[:, 0:3]; g[:, :, 1::2, 0::2] = got[:, 3:6]; g[:, :, 0::2, 1::2] = got[:, 6:9]; g[:, :, 1::2, 1::2] = got[:, 9:12]
        for b in range(3):
            exp, eratio = orc.letterbox(frames[b], S)
            assert ratio == eratio
            np.testing.assert_array_equal(g[b], exp)


def _pil_wave_cases(rng, B, H, W, MAXN):
    xyxy = np.zeros((B, MAXN, 7))
    for b in range(B):
        w = rng.uniform(3, 200, MAXN); h = rng.uniform(3, 620, MAXN)
        x = rng.uniform(-20, W - 10, MAXN); y = rng.uniform(-20, H - 10, MAXN)
        xyxy[b, :, :4] = np.stack([x, y, x + w, y + h], 1)
    xyxy[0, 0, :4] = [10, 10, 138, 266]               # exactly the target size
    xyxy[0, 1, :4] = [10, 10, 74, 138]                # exactly half: 2x up-scaling on both axes
    xyxy[0, 2, :4] = [100, 100, 260, 612]             # 160 x 512: the widest / tallest staged crop, scale 2 vertically (2-row mini-bands)
    xyxy[0, 3, :4] = [100, 100, 261, 300]             # 161 px: one too wide for the staging rows -> direct path
    xyxy[0, 4, :4] = [100, 100, 200, 613]             # scale just above 2 vertically -> direct path
    xyxy[0, 5, :4] = [W - 60, H - 200, W + 50, H + 50]      # clipped bottom-right corner
    xyxy[0, 6, :4] = [500, 500, 500.4, 500.9]         # empty -> zeros
    xyxy[0, 7, :4] = [300, 300, 303, 304]             # 3 x 4 px
    xyxy[0, 8, :4] = [300, 300, 420, 607]             # vertical scale 1.199 (4-row mini-bands, 8 source rows)
    xyxy[0, 9, :4] = [300, 300, 420, 608]             # vertical scale 1.203 (2-row mini-bands)
    xyxy[B - 1, 0, :4] = [W - 130, H - 300, W - 1, H - 1]   # the last rows of the last frame: loads end at the end of the buffer
    xyxy[B - 1, 1, :4] = [0, H - 9, 100, H - 1]
    return xyxy


@pytest.mark.parametrize("oh", [256, 384, 130])
def test_pil_wave_kernel_16bit_nhwc_equals_the_fp32_kernel_and_the_oracle(orc, oh):
    """pil_wave_kernel (r03: 128-wide NHWC 16-bit targets) against pil_crop_kernel's fp32 output (itself pinned by the Pillow-made golden and the
    oracle above) rounded to the 16-bit type, over up- and down-scaling, clipped, tiny, too-wide / too-tall (direct path), 2-row mini-band and
    end-of-buffer crops; and against the oracle directly for the first frame."""
    import torch
    from tracklab_amd import _lib
    rng = np.random.default_rng(oh)
    B, H, W, MAXN = 3, 1080, 1920, 40
    frames = _frames(rng, B, H, W)
    xyxy = _pil_wave_cases(rng, B, H, W, MAXN)
    counts = np.array([MAXN, 29, MAXN], dtype=np.int32)
    d, dx, dc = torch.from_numpy(frames).cuda(), torch.from_numpy(xyxy).cuda(), torch.from_numpy(counts).cuda()
    ref32 = _lib.roi_crop_pil_resize_norm(d, dx, dc, oh, 128, "nhwc", torch.float32)
    for dtype in (torch.float16, torch.bfloat16):
        for swap in (False, True):
            got = _lib.roi_crop_pil_resize_norm(d, dx, dc, oh, 128, "nhwc", dtype, swap_rb=swap)
            exp = _lib.roi_crop_pil_resize_norm(d, dx, dc, oh, 128, "nhwc", torch.float32, swap_rb=swap).to(dtype)
            torch.cuda.synchronize()
            assert torch.equal(got, exp), (dtype, swap, int((got != exp).sum()))
    got = ref32.cpu().numpy()                            # (B * MAXN, 3, oh, 128) logical NCHW view of NHWC memory
    for i in range(12):
        exp, _ = orc.ssort_reid_preprocess(frames[0], xyxy[0, i, :4], oh, 128)
        np.testing.assert_array_equal(got[i], exp, err_msg=f"crop {i}")
    h16 = _lib.roi_crop_pil_resize_norm(d, dx, dc, oh, 128, "nhwc", torch.float16).cpu().numpy()
    for i in range(12):
        exp, _ = orc.ssort_reid_preprocess(frames[0], xyxy[0, i, :4], oh, 128)
        np.testing.assert_array_equal(h16[i], exp.astype(np.float16), err_msg=f"crop {i} f16")


@pytest.mark.parametrize("env", [{"TLK_PIL_WAVE": "0"}, {"TLK_CROP_P16": "0"}], ids=["pil_crop_kernel", "pil_wave_kernel_general_pitch"])
def test_the_older_pil_crop_kernel_stays_bit_exact(env):
    """pil_crop_kernel stays selectable with TLK_PIL_WAVE=0, pil_wave_kernel's general-pitch code with TLK_CROP_P16=0 (the 1080p frames of the tests take
    the 16-byte-pitch specialisation otherwise); both are read once per process, hence the subprocess."""
    import subprocess
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(here, "test_gpu_image.py"), "-m", "gpu", "-q", "-x", "-k", "test_pil_ and not older"],
                       env=dict(os.environ, **env), capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and " passed" in r.stdout and "failed" not in r.stdout, r.stdout[-2000:] + r.stderr[-1000:]


@pytest.mark.parametrize("tag", ["a", "b"])
def test_pil_wave_kernel_against_pillow_over_a_sweep_of_crop_sizes(tag):
    """r03 fixture pil_sweep.npz, made by Pillow itself: 40 crops of 2 .. 200 x 2 .. 600 px (2 / 3 / 5 taps per axis, scales around 1.2 and 2, staged and
    direct crops, clipped and tiny ones, a 3 x 500 crop that Pillow's Image.resize resizes VERTICALLY first; frame widths 1280 and 1283). The 16-bit NHWC
    output (pil_wave_kernel) is mapped back to uint8 through the normalisation table and compared by SHA-256 and every 8th row; fp32 NCHW (pil_crop_kernel) too."""
    import hashlib

    import torch
    from test_oracle_motion import _pil_sweep_frame
    from tracklab_amd import _lib
    g = np.load(os.path.join(GOLDEN, "pil_sweep.npz"))
    img = _pil_sweep_frame(int(g[f"width_{tag}"]))
    n = len(g[f"boxes_{tag}"])
    boxes = torch.zeros((1, n, 7), dtype=torch.float64, device="cuda")
    boxes[0, :, :4] = torch.from_numpy(g[f"boxes_{tag}"]).cuda()
    counts = torch.tensor([n], dtype=torch.int32, device="cuda")
    d = torch.from_numpy(img).cuda()[None].contiguous()
    lut = g["norm_lut"]                                     # (3, 256) float32, strictly increasing per channel
    for dtype, layout in ((torch.float16, "nhwc"), (torch.bfloat16, "nhwc"), (torch.float32, "nchw")):
        out = _lib.roi_crop_pil_resize_norm(d, boxes, counts, 256, 128, layout, dtype).float().cpu().numpy()       # logical (n, 3, 256, 128)
        table = torch.from_numpy(lut).to(dtype).float().numpy()
        assert all(len(np.unique(table[c])) == 256 for c in range(3)) or dtype == torch.bfloat16
        for i in range(n):
            if dtype == torch.bfloat16:                    # 8 mantissa bits do not separate the 256 levels: compare the values themselves on the stored rows
                exp = np.stack([table[c][g[f"rows_{tag}"][i][:, :, c]] for c in range(3)])
                np.testing.assert_array_equal(out[i][:, ::8], exp, err_msg=f"box {i} bf16")
                continue
            u8 = np.stack([np.searchsorted(table[c], out[i, c]) for c in range(3)], axis=-1).astype(np.uint8)    # (256, 128, 3)
            np.testing.assert_array_equal(np.stack([table[c][u8[:, :, c]] for c in range(3)]), out[i], err_msg=f"box {i}: not table values")
            np.testing.assert_array_equal(u8[::8], g[f"rows_{tag}"][i], err_msg=f"box {i} {dtype} {g[f'boxes_int_{tag}'][i]}")
            assert hashlib.sha256(np.ascontiguousarray(u8).tobytes()).hexdigest() == str(g[f"sha_{tag}"][i]), f"box {i} {dtype}"
