"""-m gpu: parity in the condition the product runs in.  The fused pipeline launches association, part distance (MFMA), camera-motion estimation
and -- with several streams, or a side-stream MFMA kernel still running when the next step starts -- the crop / letterbox kernels BESIDE the
backbones' matrix kernels (reference behaviour to match is sequential and deterministic: tracklab/engine/engine.py:148-185).  r02 found a
lane-dependent result in exactly that condition; r03 named the instruction (packed FP32 arithmetic with op_sel[src1] = 1 returns
lanes 48-63 of its low half computed with src1 = 0 while MFMA kernels of another stream share the compute units) and keeps it out of the
library (tools/audit_pk_f32.py, run by build()).  This file re-runs the bit-exact parity tests of every hand-written kernel family while a
bf16 ResNet-50 forward keeps the matrix cores busy from another stream.  The host entry points launch on the legacy default stream; the load
runs on a non-blocking side stream from a helper thread.  First run on hardware: r03, 12/12 of the r02 cases; default-on since."""
import contextlib
import importlib
import inspect
import threading

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

CASES = [
    # trackers / camera-motion application (side stream in the product)
    ("test_gpu_ocsort", "test_hip_ocsort_reset_and_streams_are_independent", {}),
    ("test_gpu_ocsort", "test_hip_ocsort_device_batched_entry_point", {}),
    ("test_gpu_bytetrack", "test_bytetrack_gpu_bit_exact_vs_oracle_and_min_confidence", {}),
    ("test_gpu_bytetrack", "test_bytetrack_bank_batched_frames_and_reset", {}),
    ("test_gpu_botsort", "test_botsort_gpu_vs_oracle_and_min_confidence", {"D": 32}),
    ("test_gpu_botsort", "test_botsort_bank_batched_frames_and_reset", {}),
    ("test_gpu_deepocsort", "test_deepocsort_bank_batched_frames_and_reset", {}),
    ("test_gpu_bpbss", "test_hip_bpbss_device_batched_entry_point", {}),
    ("test_gpu_ssort", "test_plain_strongsort_gpu_kf_state_bit_exact_vs_oracle", {}),
    ("test_gpu_camera_motion", "test_plain_strongsort_camera_update_on_device", {}),
    ("test_gpu_camera_motion", "test_deepocsort_affine_correction_on_device", {}),
    ("test_gpu_camera_motion", "test_botsort_multi_gmc_on_device", {}),
    # r03: the pre/post-processing kernels (still built with packed code: they overlap MFMA kernels when a side-stream part distance / cosine
    # kernel is still running, and always with n_streams > 1)
    ("test_gpu_image", "test_letterbox_matches_oracle", {"H": 1080, "W": 1920, "layout": "focus_nhwc"}),
    ("test_gpu_image", "test_letterbox_matches_oracle", {"H": 333, "W": 517, "layout": "nchw"}),
    ("test_gpu_image", "test_crop_resize_norm_matches_oracle", {"oh": 384, "ow": 128, "layout": "nhwc"}),
    ("test_gpu_image", "test_crop_resize_norm_matches_oracle", {"oh": 256, "ow": 192, "layout": "nchw"}),
    ("test_gpu_image", "test_pil_crop_resize_norm_golden_bit_exact", {}),
    ("test_gpu_image", "test_pil_crop_resize_norm_vs_oracle_1080p", {"oh": 256, "ow": 128}),
    ("test_gpu_image", "test_yolox_decode_nms_matches_oracle", {"num_classes": 1, "nobj": 100}),
    ("test_gpu_image", "test_fused_bias_act_epilogue_matches_torch", {"dtype_name": "float16", "act": "silu"}),
    ("test_gpu_pose", "test_pose_crop_warp_norm_bit_exact", {"dtype_name": "float16", "layout": "nhwc"}),
    ("test_gpu_pose", "test_simcc_decode_matches_oracle_and_feeds_the_tracker_layout", {}),
    # r03: the MFMA kernels of the library itself, LSA, stateless filters, estimators, evaluators
    ("test_gpu_bpbss", "test_partdist_mfma_matches_oracle", {"T": 100, "N": 100, "K": 6, "D": 256}),
    ("test_gpu_kernels", "test_cosine_gallery_mfma_matches_reference_golden_and_oracle", {}),
    ("test_gpu_kernels", "test_lsa_batched_vs_scipy", {"shape": (100, 130)}),
    ("test_gpu_kernels", "test_kf8_stateless_golden_and_oracle", {}),
    ("test_gpu_kernels", "test_iou_family_golden_bit_exact", {}),
    ("test_gpu_cmc", "test_cmc_stages_equal_the_oracle", {}),
    ("test_gpu_ecc", "test_estimator_on_frames_equals_the_oracle_and_recovers_the_motion", {}),
    ("test_gpu_eval", "test_gpu_hota_on_a_full_stream_equals_the_numpy_restatement", {}),
    ("test_gpu_eval", "test_gpu_clearmot_matches_vendored_motmetrics", {}),
    ("test_gpu_setorder", "test_plain_strongsort_set_order_golden_on_gpu", {}),
]


@contextlib.contextmanager
def resnet_load():
    import torch
    from tracklab_amd.backbones.reid import part_based_reid
    reid = part_based_reid(1, 128, device="cuda", dtype=torch.bfloat16, channels_last=True)
    crops = torch.randn(96, 3, 256, 128, device="cuda", dtype=torch.bfloat16).to(memory_format=torch.channels_last)
    with torch.no_grad():
        reid(crops)
    torch.cuda.synchronize()
    stop, forwards = threading.Event(), [0]

    def run():
        torch.cuda.set_device(0)
        side = torch.cuda.Stream()
        with torch.no_grad(), torch.cuda.stream(side):
            while not stop.is_set():
                for _ in range(4):
                    reid(crops)
                side.synchronize()                      # bound the queue
                forwards[0] += 4
    th = threading.Thread(target=run, daemon=True)
    th.start()
    try:
        yield forwards
    finally:
        stop.set()
        th.join(timeout=60)
        torch.cuda.synchronize()


def _case_id(case):
    m, f, kw = case
    tail = ",".join(str(v) for v in kw.values())
    return f"{m[9:]}::{f[5:45]}" + (f"[{tail}]" if tail else "")


@pytest.mark.parametrize("module,func,kwargs", CASES, ids=[_case_id(c) for c in CASES])
def test_parity_holds_under_a_concurrent_resnet_forward(orc, module, func, kwargs):
    fn = getattr(importlib.import_module(module), func)
    if "orc" in inspect.signature(fn).parameters:
        kwargs = dict(kwargs, orc=orc)
    with resnet_load() as forwards:
        fn(**kwargs)
    assert forwards[0] > 0                               # the load really ran beside the test


def test_two_stream_fused_pipeline_ids_equal_oracle_under_load(orc):
    """DetReidTrackPipeline with n_streams = 2 (two videos per GPU: the crops of one stream's step and the part-distance / association kernels
    of the other run side by side), under the ResNet load on a third stream: both streams' ids must equal the oracle chain."""
    import torch
    from tracklab_amd import gpu_pipeline as gp
    from tracklab_amd.synth import SyntheticStream, render_frame, synth_yolox_head
    from test_gpu_pipeline_configs import _detector_rows
    S, F, steps, nobj, maxd = 2, 4, 6, 40, 48
    pipe = gp.DetReidTrackPipeline("s", n_streams=S, frames_per_step=F, max_dets=maxd, dim=64, use_graph=False)
    rng = np.random.default_rng(5)
    streams = [list(SyntheticStream(70 + s, nobj, F * steps)) for s in range(S)]
    heads = np.stack([np.stack([synth_yolox_head(rng, fr["dets"][:, :4], ratio=pipe.ratio) for fr in st]) for st in streams])   # (S, T, A, 6)
    pool = np.stack([render_frame(rng, streams[s][f]["gt_boxes"]) for s in range(S) for f in range(F)])
    d_frames = torch.from_numpy(pool).cuda()
    refs = [orc.StrongSORT(pipe.K, pipe.D, **pipe.tracker_cfg) for _ in range(S)]
    rows_total = 0
    with resnet_load() as forwards:
        for k in range(steps):
            d_heads = torch.from_numpy(np.ascontiguousarray(heads[:, k * F:(k + 1) * F].reshape(S * F, -1, 6))).cuda()
            h_rows, h_cnt = pipe.step(d_frames, d_heads)
            pipe.synchronize()
            rows, cnt = pipe.rows_numpy(h_rows, h_cnt)
            emb = pipe.last["emb"].cpu().numpy().reshape(S, F, maxd, pipe.K, pipe.D)
            vis = pipe.last["vis"].cpu().numpy().reshape(S, F, maxd, pipe.K)
            counts = pipe.last["counts"].cpu().numpy().reshape(S, F)
            for s in range(S):
                for f in range(F):
                    ltwh = _detector_rows(orc, heads[s, k * F + f], pipe.ratio)
                    n = len(ltwh)
                    assert n == int(counts[s, f])
                    ids = (k * S * F + s * F + f) * maxd + np.arange(n)       # det_id_base = frames_done * max_dets, frames stream-major
                    got = rows[s][f]
                    exp = refs[s].update(ids, ltwh.astype(np.float64), emb[s, f, :n], vis[s, f, :n], np.ones(n))
                    assert len(got) == len(exp), (k, s, f)
                    np.testing.assert_array_equal(got["det_id"], exp["det_id"], err_msg=f"step {k} stream {s} frame {f}")
                    np.testing.assert_array_equal(got["track_id"], exp["track_id"], err_msg=f"step {k} stream {s} frame {f}")
                    np.testing.assert_array_equal(got["matched_name"], exp["matched_name"])
                    rows_total += len(exp)
    assert forwards[0] > 0 and rows_total > 0.8 * S * F * steps * nobj
    pipe.close()
