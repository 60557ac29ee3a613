"""-m gpu: the backbones' compute precision. The reference runs its networks in fp32 (ONNXRuntime detector / pose, torchreid ReID;
`fp16: false` in configs/modules/track/strong_sort.yaml:10); bench.py's default IS fp32 since r04 (`value`), with the f16 and split-precision legs
reported beside it. This file states the tolerance that makes the narrower legs admissible and checks it on the same weights:

  * part embeddings (N, K, D): per-part cosine distance between the f16 and the f32 network's output <= EMB_COS_TOL, and the
    part-based ReID distance matrix the tracker consumes (tlk_partdist_f32) moves by <= DIST_TOL (its gate `max_dist` is 0.5);
  * visibility flags identical;
  * the tracks: detection -> track id assignment of the fused pipeline over a 48-frame stream is IDENTICAL under f16 and f32.
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

EMB_COS_TOL = 1e-5      # measured 3.6e-7 on the random-init R50 (printed below)
DIST_TOL = 2e-3         # absolute, on distances in [0, 1] (measured 1.9e-4); the association gate max_dist is 0.5


def _crops(dtype, n=64):
    import torch
    from tracklab_amd import _lib
    from tracklab_amd.synth import SyntheticStream, render_frame
    fr = SyntheticStream(5, n, 1).step()
    frame = render_frame(np.random.default_rng(1), fr["gt_boxes"])
    d = fr["dets"]
    boxes = np.zeros((1, 128, 4), np.float32)
    boxes[0, :len(d)] = np.column_stack([d[:, 0], d[:, 1], d[:, 2] - d[:, 0], d[:, 3] - d[:, 1]])
    c = _lib.roi_crop_resize_norm(torch.from_numpy(frame[None]).cuda(), torch.from_numpy(boxes).cuda(),
                                  torch.tensor([len(d)], dtype=torch.int32).cuda(), 384, 128, "nhwc", dtype)
    return c[:len(d)]


def test_f16_embeddings_within_stated_tolerance_of_fp32():
    import torch
    from tracklab_amd import _lib
    from tracklab_amd.backbones.reid import part_based_reid
    K, D = 6, 256
    with torch.no_grad():
        e16, v16 = part_based_reid(K, D, dtype=torch.float16)(_crops(torch.float16))
        e32, v32 = part_based_reid(K, D, dtype=torch.float32)(_crops(torch.float32))       # same seed -> same weights
    assert torch.equal(v16, v32), "visibility flags differ between f16 and f32"
    a, b = torch.nn.functional.normalize(e16.float(), dim=-1), torch.nn.functional.normalize(e32.float(), dim=-1)
    cos_dist = (1 - (a * b).sum(-1)).abs().max().item()
    u16, u32 = v16.to(torch.uint8).contiguous(), v32.to(torch.uint8).contiguous()
    d16 = _lib.partdist(e16.float().contiguous(), u16, e16.float().contiguous(), u16)
    d32 = _lib.partdist(e32.float().contiguous(), u32, e32.float().contiguous(), u32)
    dist_err = (d16 - d32).abs().max().item()
    print(f"f16 vs f32 part embeddings: max cosine distance {cos_dist:.2e}, max |delta part distance| {dist_err:.2e}")
    assert cos_dist <= EMB_COS_TOL and dist_err <= DIST_TOL


def test_track_ids_identical_under_f16_and_f32_backbones():
    import torch
    from tracklab_amd import gpu_pipeline as gp
    from tracklab_amd.synth import SyntheticStream, render_frame, synth_yolox_head
    F, steps, nobj = 8, 6, 40
    rng = np.random.default_rng(4)
    heads, frames = [], []
    stream = list(SyntheticStream(17, nobj, F * steps, miss_prob=0.05))
    ratio = min(640 / 1080, 640 / 1920)
    for fr in stream:
        heads.append(synth_yolox_head(rng, fr["dets"][:, :4], ratio=ratio))
        frames.append(render_frame(rng, fr["gt_boxes"]))
    d_heads = torch.from_numpy(np.stack(heads)).cuda().reshape(steps, F, -1, 6)
    d_frames = torch.from_numpy(np.stack(frames)).cuda().reshape(steps, F, 1080, 1920, 3)
    out = {}
    for dt in (torch.float16, torch.float32):
        pipe = gp.DetReidTrackPipeline("s", n_streams=1, frames_per_step=F, max_dets=64, dim=64, dtype=dt, use_graph=False)
        rows_all = []
        for k in range(steps):
            h_rows, h_cnt = pipe.step(d_frames[k], d_heads[k])
            pipe.synchronize()
            rows, _ = pipe.rows_numpy(h_rows, h_cnt)
            rows_all += [np.stack([r["det_id"], r["track_id"]], 1) for r in rows[0]]
        out[dt] = rows_all
        pipe.close()
    assert sum(len(r) for r in out[torch.float16]) > 0.9 * nobj * F * steps * 0.95
    for f, (a, b) in enumerate(zip(out[torch.float16], out[torch.float32])):
        np.testing.assert_array_equal(a, b, err_msg=f"frame {f}: f16 and f32 backbones give different tracks")


def _scaled_reid(dtype, scale, split=False, heavy_tail=False):
    """the random-init part-based ReID net with its stem weights multiplied by `scale` (ReLU networks are positively homogeneous up to their
    biases: every activation of the backbone grows by about that factor) and, for heavy_tail, the per-channel scales a BatchNorm fold leaves behind
    drawn log-uniform over four decades"""
    import torch
    from tracklab_amd.backbones.reid import part_based_reid
    m = part_based_reid(6, 256, dtype=torch.float32, split_precision=False)
    with torch.no_grad():
        stem = next(mod for mod in m.backbone.modules() if hasattr(mod, "conv") and mod.conv.in_channels == 3)
        stem.conv.weight.mul_(scale)
        stem.bias.mul_(scale)
        # ... and the 1 x 1 reduction behind the backbone undoes it, so the HEAD sees the natural scale: scaled logits would turn the part softmax
        # into an arg-max and amplify a 1e-3 relative difference into flipped attention maps -- a temperature effect, not a precision one
        m.reduce.conv.weight.div_(scale)
        if heavy_tail:
            # per-channel scales log-uniform over four decades on every bottleneck's 3 x 3 output, undone on the input channels of the 1 x 1
            # expansion that follows (ReLU commutes with a positive scale): the FUNCTION is unchanged, the intermediate tensors are heavy-tailed
            g = torch.Generator(device="cpu").manual_seed(3)
            for blk in m.backbone.modules():
                if type(blk).__name__ == "_Bottleneck":
                    f = torch.exp(torch.empty(blk.c2.conv.out_channels).uniform_(-4.6, 4.6, generator=g)).to(blk.c2.conv.weight.device)      # 1e-2 .. 1e2
                    blk.c2.conv.weight.mul_(f[:, None, None, None]); blk.c2.bias.mul_(f)
                    blk.c3.conv.weight.div_(f[None, :, None, None])
    if dtype == torch.float16:
        m = m.half()
    m.split_precision = bool(split)
    return m


def test_precision_envelope_of_the_f16_and_split_legs():
    """WHERE do the narrower legs stop being admissible?  The stem of the (random-init) ReID net is scaled so that
    the LARGEST activation anywhere in the backbone (every convolution's output, fp32 run) reaches ~1e3 ... ~3e5; per scale: that maximum, the
    f16 and split-precision embeddings against the exact-fp32 run's (max per-part cosine distance), finiteness.  Stated envelope (DESIGN.md
    section 2, from this table):
      * f16:   cosine distance <= 1e-5 while every activation stays inside float16's range (relative error is scale-free until the range ends
               at 65504); beyond it an activation saturates to infinity and the embeddings are NOT finite;
      * split: fp32-class (<= 1e-6) over the same range AND beyond it (r06: the (hi, lo) planes carry a power-of-two scale that follows every
               layer's recorded maximum; r05 saturated at the same 65504);
      * past the range nothing is silently wrong: ReLU lets NaN through (r05: it used to turn the NaN of inf - inf into 0, and a saturated net
        produced finite garbage) and the pipeline raises TlkError (gpu_pipeline: check_finite) instead of tracking on infinities."""
    import json
    import torch
    from tracklab_amd.backbones.common import ConvBiasAct
    x32 = _crops(torch.float32, 24)
    x16 = x32.half()

    def run(m, x, track=False):
        peak = [0.0]
        hooks = []
        if track:
            def hook(mod, i, o):
                t = o.hi if hasattr(o, "hi") else o
                peak[0] = max(peak[0], float(t.detach().float().abs().max()))
            hooks = [mod.register_forward_hook(hook) for mod in m.backbone.modules() if isinstance(mod, ConvBiasAct)]
        try:
            with torch.no_grad():
                e, v = m(x)
        finally:
            for h in hooks:
                h.remove()
        return e.float(), peak[0]

    def cosd(e, ref):
        if not torch.isfinite(e).all():
            return float("inf")
        a, b = torch.nn.functional.normalize(e, dim=-1), torch.nn.functional.normalize(ref, dim=-1)
        return (1 - (a * b).sum(-1)).abs().max().item()

    base_peak = run(_scaled_reid(torch.float32, 1.0), x32, track=True)[1]
    table = []
    for target in (1e3, 1e4, 3e4, 5e4, 1.5e5, 4e5):
        sc = target / base_peak
        e32, a32 = run(_scaled_reid(torch.float32, sc), x32, track=True)
        e16, _ = run(_scaled_reid(torch.float16, sc), x16)
        esp, _ = run(_scaled_reid(torch.float32, sc, split=True), x32)
        table.append({"max_activation_f32": a32, "f16_cos": cosd(e16, e32), "split_cos": cosd(esp, e32), "f32_finite": bool(torch.isfinite(e32).all())})
    # heavy-tailed intermediate tensors at the natural scale: channels of one tensor spread over four decades (what BatchNorm folding can leave)
    e32h, a32h = run(_scaled_reid(torch.float32, 1.0, heavy_tail=True), x32, track=True)
    e16h, _ = run(_scaled_reid(torch.float16, 1.0, heavy_tail=True), x16)
    esph, _ = run(_scaled_reid(torch.float32, 1.0, split=True, heavy_tail=True), x32)
    heavy = {"heavy_tailed_channels": True, "max_activation_f32": a32h, "f16_cos": cosd(e16h, e32h), "split_cos": cosd(esph, e32h)}
    print("precision envelope, heavy-tailed per-channel scales (1e-2 .. 1e2 inside every bottleneck):", json.dumps(heavy))
    assert heavy["f16_cos"] <= 1e-4 and heavy["split_cos"] <= 1e-6, heavy
    print("precision envelope (largest |activation| of the backbone, f16 cosine distance, split cosine distance), natural peak %.3g:" % base_peak)
    print(json.dumps(table))
    inside = [t for t in table if t["max_activation_f32"] <= 5.5e4]
    outside = [t for t in table if t["max_activation_f32"] >= 1.2e5]
    assert len(inside) >= 4 and len(outside) >= 2
    assert all(t["f32_finite"] for t in table)
    # (at the 5e4 point the test's own 1 / scale on the reduction weights pushes them into float16's subnormals: 1.2e-5 there is the experiment, not the range)
    assert all(t["f16_cos"] <= (EMB_COS_TOL if t["max_activation_f32"] <= 3.3e4 else 2e-5) and t["split_cos"] <= 1e-6 for t in inside), table
    assert all(t["f16_cos"] == float("inf") for t in outside), "past float16's range the f16 output must be NON-finite, never finite garbage"
    # r06: the split leg carries a power-of-two scale with its (hi, lo) planes (common.SplitScales, tlk_conv2d_nhwc_16s), so it
    # no longer ends at 65504: fp32-class past float16's range too
    assert all(t["split_cos"] <= 1e-6 for t in outside), table


def test_saturated_embeddings_fail_loudly_in_the_pipeline():
    """past the envelope the fused pipeline raises instead of associating on infinities"""
    import torch
    from tracklab_amd import _lib
    from tracklab_amd import gpu_pipeline as gp
    from tracklab_amd.synth import SyntheticStream, render_frame, synth_yolox_head
    rng = np.random.default_rng(2)
    fr = SyntheticStream(3, 20, 1).step()
    ratio = min(640 / 1080, 640 / 1920)
    head = torch.from_numpy(synth_yolox_head(rng, fr["dets"][:, :4], ratio=ratio)[None]).cuda()
    frame = torch.from_numpy(render_frame(rng, fr["gt_boxes"])[None]).cuda()
    pipe = gp.DetReidTrackPipeline("s", n_streams=1, frames_per_step=1, max_dets=32, dim=64, dtype=torch.float16, use_graph=False)
    try:
        pipe.step(frame, head)
        pipe.synchronize()                                   # in range: fine
        with torch.no_grad():
            stem = next(mod for mod in pipe.reid.backbone.modules() if hasattr(mod, "conv") and mod.conv.in_channels == 3)
            stem.conv.weight.mul_(1e5)
        pipe.step(frame, head)
        with pytest.raises(_lib.TlkError, match="not finite"):
            pipe.synchronize()
        pipe.reset()
    finally:
        pipe.close()
