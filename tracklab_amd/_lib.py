"""ctypes binding of libtlk.so (``include/tlk.h``).

There is NO CPU fallback: if the shared library is missing, or no gfx950 device is visible when a
handle is created, this raises. (The oracle under ``oracle/`` is test infrastructure and is never
imported from here.)
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# TLK_LIB_PATH: another build of the same library (tests/test_gpu_canary.py loads the guard-word build, tools/build_canary.sh)
LIB_PATH = os.environ.get("TLK_LIB_PATH") or os.path.join(_HERE, "lib", "libtlk.so")

ASSO = {"iou": 0, "giou": 1, "diou": 2, "ciou": 3, "ct_dist": 4}

_dp = C.POINTER(C.c_double)
_ip32 = C.POINTER(C.c_int32)
_ip64 = C.POINTER(C.c_int64)


class TlkError(RuntimeError):
    pass


class OcsortParams(C.Structure):
    _fields_ = [("det_thresh", C.c_double), ("iou_threshold", C.c_double), ("inertia", C.c_double),
                ("min_confidence", C.c_double),
                ("max_age", C.c_int32), ("min_hits", C.c_int32), ("delta_t", C.c_int32),
                ("asso_func", C.c_int32), ("use_byte", C.c_int32), ("wrapper_mode", C.c_int32),
                ("max_tracks", C.c_int32), ("max_dets", C.c_int32)]


_lib = None


def build() -> str:
    """Compile every HIP source for gfx950 into tracklab_amd/lib/libtlk.so (hipcc cross-compiles
    without a GPU)."""
    import subprocess
    subprocess.run(["bash", os.path.join(_HERE, "csrc", "build.sh")], check=True)
    return LIB_PATH


def lib():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise TlkError(f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                       "(tracklab_amd has no CPU fallback)")
    try:
        # One HIP runtime per process: PyTorch-ROCm bundles its own libamdhip64 (SONAME libamdhip64.so.7). Loading it
        # first lets libtlk's DT_NEEDED resolve to the same copy; the other order loads /opt/rocm's copy next to torch's
        # and torch then reports "No HIP GPUs are available".
        import torch  # noqa: F401
    except ImportError:
        pass
    L = C.CDLL(LIB_PATH)
    L.tlk_last_error.restype = C.c_char_p
    L.tlk_version.restype = C.c_int
    L.tlk_device_count.argtypes = [C.POINTER(C.c_int)]
    L.tlk_iou_matrix_f64.argtypes = [C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
    L.tlk_lsa_f64.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    L.tlk_ocsort_create.argtypes = [C.POINTER(OcsortParams), C.c_int, C.c_int, C.POINTER(C.c_void_p)]
    L.tlk_ocsort_destroy.argtypes = [C.c_void_p]
    L.tlk_ocsort_reset.argtypes = [C.c_void_p, C.c_int]
    L.tlk_ocsort_update.argtypes = [C.c_void_p, C.c_int, _dp, C.c_int, _dp, C.c_int, C.POINTER(C.c_int)]
    L.tlk_ocsort_update_dev.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int,
                                        C.c_void_p, C.c_void_p]
    L.tlk_ocsort_get_tracks.argtypes = [C.c_void_p, C.c_int, _dp, _dp, _ip64, C.c_int, C.POINTER(C.c_int)]
    _lib = L
    return L


def check(code: int) -> None:
    if code != 0:
        raise TlkError(f"libtlk error {code}: {lib().tlk_last_error().decode()}")


def device_count() -> int:
    n = C.c_int(0)
    lib().tlk_device_count(C.byref(n))
    return n.value


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def current_stream_ptr():
    """hipStream_t of torch's current stream (0 if torch is not imported / no GPU)."""
    import torch
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


class OCSortBank:
    """``n_streams`` device-resident OC-SORT trackers (``tlk_ocsort_*``).

    Mirrors ``oc_sort.ocsort.OCSort`` hyper-parameters (ocsort.py:186-201); ``wrapper_mode`` adds the
    per-frame behaviour of ``OCSORT.process`` (oc_sort_api.py:50-56).
    """

    def __init__(self, det_thresh, max_age=30, min_hits=3, iou_threshold=0.3, delta_t=3,
                 asso_func="iou", inertia=0.2, use_byte=False, *, min_confidence=0.0,
                 wrapper_mode=False, n_streams=1, device=0, max_tracks=256, max_dets=128):
        if asso_func not in ASSO:
            raise KeyError(asso_func)
        self.params = OcsortParams(det_thresh, iou_threshold, inertia, min_confidence, max_age, min_hits,
                                   delta_t, ASSO[asso_func], int(use_byte), int(wrapper_mode),
                                   max_tracks, max_dets)
        self.n_streams, self.device = n_streams, device
        self.max_tracks, self.max_dets = max_tracks, max_dets
        h = C.c_void_p()
        check(lib().tlk_ocsort_create(C.byref(self.params), n_streams, device, C.byref(h)))
        self._h = h
        self._out = np.empty((max_tracks + max_dets, 8))

    def close(self):
        if getattr(self, "_h", None):
            lib().tlk_ocsort_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def reset(self, stream: int = -1):
        check(lib().tlk_ocsort_reset(self._h, stream))

    def update(self, dets, stream: int = 0) -> np.ndarray:
        dets = _f64(dets).reshape(-1, 7)
        n = C.c_int(0)
        check(lib().tlk_ocsort_update(self._h, stream, dets.ctypes.data_as(_dp), len(dets),
                                      self._out.ctypes.data_as(_dp), len(self._out), C.byref(n)))
        return self._out[:n.value].copy()

    def update_dev(self, dets_ptr, counts_ptr, n_frames, out_ptr, out_cap, out_counts_ptr, stream_ptr=None):
        check(lib().tlk_ocsort_update_dev(self._h, dets_ptr, counts_ptr, n_frames, out_ptr, out_cap,
                                          out_counts_ptr, stream_ptr))

    def tracks(self, stream: int = 0):
        cap = self.max_tracks
        x, P, ids = np.empty((cap, 7)), np.empty((cap, 7, 7)), np.empty(cap, dtype=np.int64)
        n = C.c_int(0)
        check(lib().tlk_ocsort_get_tracks(self._h, stream, x.ctypes.data_as(_dp), P.ctypes.data_as(_dp),
                                          ids.ctypes.data_as(_ip64), cap, C.byref(n)))
        return x[:n.value], P[:n.value], ids[:n.value]


# ------------------------------------------------------------------------------------------------
# image ops on torch device tensors (torch is plumbing here: device memory + the current stream)
# ------------------------------------------------------------------------------------------------
LAYOUT = {"nchw": 0, "nhwc": 1, "focus_nhwc": 2}
IMAGENET_MEAN = (0.485, 0.456, 0.406)
IMAGENET_STD = (0.229, 0.224, 0.225)


def _dtype_code(dtype):
    import torch
    return {torch.float32: 0, torch.float16: 1, torch.bfloat16: 2}[dtype]


def _bind_image(L):
    if getattr(L, "_image_bound", False):
        return
    L.tlk_letterbox_u8.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p,
                                   C.POINTER(C.c_double), C.c_void_p]
    L.tlk_roi_crop_resize_norm.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int,
                                           C.c_int, C.c_int, C.POINTER(C.c_float), C.POINTER(C.c_float), C.c_int,
                                           C.c_int, C.c_void_p, C.c_void_p]
    L.tlk_roi_crop_resize_norm_compact.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int,
                                                   C.c_int, C.c_int, C.POINTER(C.c_float), C.POINTER(C.c_float), C.c_int,
                                                   C.c_int, C.c_void_p, C.c_void_p]
    L.tlk_crop_slot_bases.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    L.tlk_conv_set_dynamic_batch.argtypes = [C.c_void_p]
    L.tlk_yolox_decode_nms.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_float, C.c_float, C.c_float,
                                       C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                       C.c_void_p, C.c_void_p, C.c_int64, C.c_double, C.c_void_p]
    L._image_bound = True


SWAP_RB = 0x100     # include/tlk.h TLK_SWAP_RB: output channel c = source channel 2 - c


def letterbox(frames, size=640, layout="nchw", dtype=None, out=None, swap_rb=False):
    """frames: (B,H,W,3) uint8 cuda tensor -> letterboxed tensor (logical NCHW shape; memory per `layout`)
    and the resize ratio (rtmlib YOLOX.preprocess). swap_rb: read the (RGB) frames as BGR, like the reference's detector does."""
    import torch
    L = lib()
    _bind_image(L)
    assert frames.is_cuda and frames.dtype == torch.uint8 and frames.is_contiguous() and frames.dim() == 4
    dtype = dtype or torch.float16
    B, H, W, _ = frames.shape
    if out is None:
        if layout == "nchw":
            out = torch.empty((B, 3, size, size), dtype=dtype, device=frames.device)
        elif layout == "nhwc":
            out = torch.empty((B, size, size, 3), dtype=dtype, device=frames.device)
        else:
            out = torch.empty((B, size // 2, size // 2, 12), dtype=dtype, device=frames.device)
    ratio = C.c_double(0)
    check(L.tlk_letterbox_u8(frames.data_ptr(), B, H, W, size, LAYOUT[layout] | (SWAP_RB if swap_rb else 0), _dtype_code(dtype), out.data_ptr(),
                             C.byref(ratio), current_stream_ptr()))
    if layout != "nchw":
        out = out.permute(0, 3, 1, 2)        # logical NCHW view of channels-last memory
    return out, ratio.value


def crop_slot_bases(counts, max_n, base=None, total=None, slot_of=None):
    """Bookkeeping of a DENSE crop batch (``tlk_crop_slot_bases``): counts (B,) i32 -> base (B,) i32 exclusive prefix sums, total (1,) i32
    = the number of real crops, slot_of (B*max_n,) i64 = position of crop i of frame b in the dense batch (a valid position for padding
    slots too).  All on the device, in stream order: nothing here needs the host to know the counts."""
    import torch
    L = lib()
    _bind_image(L)
    assert counts.is_cuda and counts.dtype == torch.int32 and counts.is_contiguous()
    B = counts.numel()
    base = torch.empty(B, dtype=torch.int32, device=counts.device) if base is None else base
    total = torch.empty(1, dtype=torch.int32, device=counts.device) if total is None else total
    slot_of = torch.empty(B * max_n, dtype=torch.int64, device=counts.device) if slot_of is None else slot_of
    assert base.dtype == torch.int32 and total.dtype == torch.int32 and slot_of.dtype == torch.int64 and slot_of.numel() == B * max_n
    check(L.tlk_crop_slot_bases(counts.data_ptr(), B, max_n, base.data_ptr(), total.data_ptr(), slot_of.data_ptr(), current_stream_ptr()))
    return base, total, slot_of


def conv_set_dynamic_batch(n_images=None):
    """``tlk_conv_set_dynamic_batch``: every libtlk convolution launched (or captured into a hipGraph) from now on reads its image count
    from ``n_images[0]`` (1-element int32 CUDA tensor, kept alive by the caller) when the kernel runs; None switches it off."""
    import torch
    L = lib()
    _bind_image(L)
    if n_images is None:
        check(L.tlk_conv_set_dynamic_batch(None))
        return
    assert n_images.is_cuda and n_images.dtype == torch.int32 and n_images.numel() >= 1
    check(L.tlk_conv_set_dynamic_batch(n_images.data_ptr()))


def roi_crop_resize_norm(frames, boxes_ltwh, counts, out_h, out_w, layout="nchw", dtype=None,
                         mean=IMAGENET_MEAN, std=IMAGENET_STD, out=None, swap_rb=False, slot_base=None):
    """frames (B,H,W,3) u8, boxes_ltwh (B,max_n,4) f32, counts (B,) i32 -> (B*max_n, 3, out_h, out_w).
    slot_base (B,) i32 (``crop_slot_bases``): write the crops as a dense batch -- crop i of frame b at slot slot_base[b] + i."""
    import torch
    L = lib()
    _bind_image(L)
    assert frames.is_cuda and frames.dtype == torch.uint8 and frames.is_contiguous()
    assert boxes_ltwh.dtype == torch.float32 and boxes_ltwh.is_contiguous() and counts.dtype == torch.int32
    dtype = dtype or torch.float16
    B, H, W, _ = frames.shape
    max_n = boxes_ltwh.shape[1]
    if out is None:      # zeros: the kernel does not touch the padding slots (i >= counts[b])
        shape = (B * max_n, 3, out_h, out_w) if layout == "nchw" else (B * max_n, out_h, out_w, 3)
        out = torch.zeros(shape, dtype=dtype, device=frames.device)
    m = (C.c_float * 3)(*mean)
    s = (C.c_float * 3)(*std)
    if slot_base is not None:
        assert slot_base.is_cuda and slot_base.dtype == torch.int32 and slot_base.numel() == B
        check(L.tlk_roi_crop_resize_norm_compact(frames.data_ptr(), B, H, W, boxes_ltwh.data_ptr(), counts.data_ptr(), slot_base.data_ptr(), max_n,
                                                 out_h, out_w, m, s, LAYOUT[layout] | (SWAP_RB if swap_rb else 0), _dtype_code(dtype), out.data_ptr(),
                                                 current_stream_ptr()))
    else:
        check(L.tlk_roi_crop_resize_norm(frames.data_ptr(), B, H, W, boxes_ltwh.data_ptr(), counts.data_ptr(), max_n,
                                         out_h, out_w, m, s, LAYOUT[layout] | (SWAP_RB if swap_rb else 0), _dtype_code(dtype), out.data_ptr(),
                                         current_stream_ptr()))
    if layout != "nchw":
        out = out.permute(0, 3, 1, 2)
    return out


def roi_crop_pil_resize_norm(frames, boxes_xyxy, counts, out_h=256, out_w=128, layout="nchw", dtype=None,
                             mean=IMAGENET_MEAN, std=IMAGENET_STD, out=None, swap_rb=False):
    """Plain StrongSORT's ReID input (int-truncated crop + Pillow bilinear resize + ToTensor + Normalize).
    frames (B,H,W,3) u8, boxes_xyxy (B,max_n,S>=4) f64 whose first four columns are x1,y1,x2,y2 (the tracker's (n,7)
    detection rows can be passed as they are), counts (B,) i32 -> (B*max_n, 3, out_h, out_w)."""
    import torch
    L = lib()
    _bind_image(L)
    if not getattr(L, "_pil_bound", False):
        L.tlk_roi_crop_pil_resize_norm.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int,
                                                   C.c_int, C.c_int, C.POINTER(C.c_float), C.POINTER(C.c_float), C.c_int, C.c_int,
                                                   C.c_void_p, C.c_void_p]
        L._pil_bound = True
    assert frames.is_cuda and frames.dtype == torch.uint8 and frames.is_contiguous()
    assert boxes_xyxy.dtype == torch.float64 and boxes_xyxy.is_contiguous() and boxes_xyxy.dim() == 3 and counts.dtype == torch.int32
    dtype = dtype or torch.float16
    B, H, W, _ = frames.shape
    max_n, stride = boxes_xyxy.shape[1], boxes_xyxy.shape[2]
    if out is None:      # zeros: the kernel does not touch the padding slots (i >= counts[b])
        shape = (B * max_n, 3, out_h, out_w) if layout == "nchw" else (B * max_n, out_h, out_w, 3)
        out = torch.zeros(shape, dtype=dtype, device=frames.device)
    m = (C.c_float * 3)(*mean)
    s = (C.c_float * 3)(*std)
    check(L.tlk_roi_crop_pil_resize_norm(frames.data_ptr(), B, H, W, boxes_xyxy.data_ptr(), stride, counts.data_ptr(), max_n,
                                         out_h, out_w, m, s, LAYOUT[layout] | (SWAP_RB if swap_rb else 0), _dtype_code(dtype), out.data_ptr(), current_stream_ptr()))
    if layout != "nchw":
        out = out.permute(0, 3, 1, 2)
    return out


RTMPOSE_MEAN = (123.675, 116.28, 103.53)
RTMPOSE_STD = (58.395, 57.12, 57.375)


def _bind_pose(L):
    if getattr(L, "_pose_bound", False):
        return
    vp, ci, cd = C.c_void_p, C.c_int, C.c_double
    L.tlk_pose_crop_warp_norm.argtypes = [vp, ci, ci, ci, vp, ci, vp, ci, ci, ci, C.POINTER(cd), C.POINTER(cd), ci, ci, vp, vp, vp]
    L.tlk_simcc_decode.argtypes = [vp, vp, ci, ci, ci, ci, cd, vp, ci, ci, vp, vp, vp, vp]
    L._pose_bound = True


def pose_crop_warp_norm(frames, boxes_xyxy, counts, in_w=192, in_h=256, layout="nchw", dtype=None, mean=RTMPOSE_MEAN,
                        std=RTMPOSE_STD, out=None, meta=None, swap_rb=False):
    """rtmlib RTMPose.preprocess for every box of a batch of frames. frames (B,H,W,3) u8, boxes_xyxy (B,max_n,S>=4) f64,
    counts (B,) i32 -> (crops (B*max_n, 3, in_h, in_w), meta (B*max_n, 10) f64 [centre, scale, inverse matrix])."""
    import torch
    L = lib()
    _bind_pose(L)
    assert frames.is_cuda and frames.dtype == torch.uint8 and frames.is_contiguous()
    assert boxes_xyxy.dtype == torch.float64 and boxes_xyxy.is_contiguous() and boxes_xyxy.dim() == 3 and counts.dtype == torch.int32
    dtype = dtype or torch.float16
    B, H, W, _ = frames.shape
    max_n, stride = boxes_xyxy.shape[1], boxes_xyxy.shape[2]
    if out is None:
        shape = (B * max_n, 3, in_h, in_w) if layout == "nchw" else (B * max_n, in_h, in_w, 3)
        out = torch.empty(shape, dtype=dtype, device=frames.device)
    if meta is None:
        meta = torch.empty((B * max_n, 10), dtype=torch.float64, device=frames.device)
    m, s = (C.c_double * 3)(*mean), (C.c_double * 3)(*std)
    check(L.tlk_pose_crop_warp_norm(frames.data_ptr(), B, H, W, boxes_xyxy.data_ptr(), stride, counts.data_ptr(), max_n, in_w, in_h,
                                    m, s, LAYOUT[layout] | (SWAP_RB if swap_rb else 0), _dtype_code(dtype), out.data_ptr(), meta.data_ptr(), current_stream_ptr()))
    if layout != "nchw":
        out = out.permute(0, 3, 1, 2)
    return out, meta


def simcc_decode(simcc_x, simcc_y, meta, in_w=192, in_h=256, split_ratio=2.0, out=None):
    """simcc_x (n,K,Wx), simcc_y (n,K,Wy) f32 cuda + meta (n,10) -> dict kps_xyc (n,K,3) f64, scores (n,K) f32, conf (n,) f32."""
    import torch
    L = lib()
    _bind_pose(L)
    assert simcc_x.is_cuda and simcc_x.dtype == torch.float32 and simcc_x.is_contiguous() and simcc_y.is_contiguous()
    n, K, Wx = simcc_x.shape
    Wy = simcc_y.shape[2]
    if out is None:
        out = {"kps_xyc": torch.empty((n, K, 3), dtype=torch.float64, device=simcc_x.device),
               "scores": torch.empty((n, K), dtype=torch.float32, device=simcc_x.device),
               "conf": torch.empty((n,), dtype=torch.float32, device=simcc_x.device)}
    check(L.tlk_simcc_decode(simcc_x.data_ptr(), simcc_y.data_ptr(), n, K, Wx, Wy, float(split_ratio), meta.data_ptr(), in_w, in_h,
                             out["kps_xyc"].data_ptr(), out["scores"].data_ptr(), out["conf"].data_ptr(), current_stream_ptr()))
    return out


def yolox_decode_nms(pred, size, ratio, img_w, img_h, max_out=128, nms_thr=0.45, score_thr=0.7,
                     out=None, trk_in=None, det_id_base=0, category_id=1.0):
    """pred (B, A, 5+C) f32 cuda -> dict of ltwh (B,max_out,4), xyxy, scores, cls, counts (rtmlib order)."""
    import torch
    L = lib()
    _bind_image(L)
    assert pred.is_cuda and pred.dtype == torch.float32 and pred.is_contiguous() and pred.dim() == 3
    B, A, F = pred.shape
    dev = pred.device
    if out is None:
        out = {"ltwh": torch.zeros((B, max_out, 4), dtype=torch.float32, device=dev),
               "xyxy": torch.zeros((B, max_out, 4), dtype=torch.float32, device=dev),
               "scores": torch.zeros((B, max_out), dtype=torch.float32, device=dev),
               "cls": torch.zeros((B, max_out), dtype=torch.int32, device=dev),
               "counts": torch.zeros((B,), dtype=torch.int32, device=dev)}
    check(L.tlk_yolox_decode_nms(pred.data_ptr(), B, size, F - 5, ratio, nms_thr, score_thr, img_w, img_h, max_out,
                                 out["ltwh"].data_ptr(), out["xyxy"].data_ptr(), out["scores"].data_ptr(),
                                 out["cls"].data_ptr(), out["counts"].data_ptr(),
                                 trk_in.data_ptr() if trk_in is not None else None, det_id_base, category_id,
                                 current_stream_ptr()))
    return out


# ------------------------------------------------------------------------------------------------
# BPBReID-StrongSORT bank + part-based distance
# ------------------------------------------------------------------------------------------------
class BpbssParams(C.Structure):
    _fields_ = [(n, C.c_double) for n in ("ema_alpha", "mc_lambda", "max_dist", "max_iou_distance", "min_bbox_confidence",
                                          "gating_thres_factor", "w_kfgd", "w_reid", "w_st")] + \
               [(n, C.c_int32) for n in ("max_age", "n_init", "only_position_for_kf_gating",
                                         "max_kalman_prediction_without_update", "matching_strategy", "wrapper_mode",
                                         "parts", "dim", "max_tracks", "max_dets", "motion_criterium", "reserved_")] + \
               [("max_oks_distance", C.c_double)]


BPBSS_ROW = np.dtype([("det_id", "<i8"), ("track_id", "<i8"), ("kf_ltwh", "<f8", (4,)), ("pred_ltwh", "<f8", (4,)),
                      ("pred_valid", "<i4"), ("matched_name", "<i4"), ("matched_dist", "<f8"), ("hits", "<i4"),
                      ("age", "<i4"), ("tsu", "<i4"), ("state", "<i4")], align=True)
MATCHING = {"strong_sort_matching": 0, "bot_sort_matching": 1}


def _bind_bpbss(L):
    if getattr(L, "_bpbss_bound", False):
        return
    L.tlk_bpbss_create.argtypes = [C.POINTER(BpbssParams), C.c_int, C.c_int, C.POINTER(C.c_void_p)]
    L.tlk_bpbss_destroy.argtypes = [C.c_void_p]
    L.tlk_bpbss_reset.argtypes = [C.c_void_p, C.c_int]
    L.tlk_bpbss_update.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                   C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.POINTER(C.c_int)]
    L.tlk_bpbss_update_dev.argtypes = [C.c_void_p] + [C.c_void_p] * 7 + [C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
    L.tlk_bpbss_get_tracks.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                       C.c_int, C.POINTER(C.c_int)]
    L.tlk_partdist_f32.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int,
                                   C.c_void_p, C.c_void_p]
    L._bpbss_bound = True


class BpbssBank:
    """``n_streams`` device-resident BPBReID-StrongSORT trackers (``tlk_bpbss_*``); hyper-parameter names follow
    ``StrongSORT.__init__`` (bpbreid_strong_sort/strong_sort.py:12-31)."""

    def __init__(self, parts, dim, ema_alpha=0.9, mc_lambda=0.995, max_dist=0.2, motion_criterium="iou",
                 max_iou_distance=0.7, max_oks_distance=0.7, max_age=30, n_init=3, nn_budget=100,
                 min_bbox_confidence=0.2, only_position_for_kf_gating=False, max_kalman_prediction_without_update=7,
                 matching_strategy="strong_sort_matching", gating_thres_factor=1.5, w_kfgd=1, w_reid=1, w_st=1, *,
                 wrapper_mode=False, n_streams=1, device=0, max_tracks=256, max_dets=128):
        if motion_criterium not in ("iou", "oks"):
            raise ValueError("motion_criterium should be either 'iou' or 'oks'")
        L = lib()
        _bind_bpbss(L)
        self.params = BpbssParams(ema_alpha, mc_lambda, max_dist, max_iou_distance, min_bbox_confidence,
                                  gating_thres_factor, w_kfgd, w_reid, w_st, max_age, n_init,
                                  int(only_position_for_kf_gating), max_kalman_prediction_without_update,
                                  MATCHING[matching_strategy], int(wrapper_mode), parts, dim, max_tracks, max_dets,
                                  {"iou": 0, "oks": 1}[motion_criterium], 0, max_oks_distance)
        self.K, self.D, self.n_streams, self.max_tracks, self.max_dets = parts, dim, n_streams, max_tracks, max_dets
        h = C.c_void_p()
        check(L.tlk_bpbss_create(C.byref(self.params), n_streams, device, C.byref(h)))
        self._h = h
        self._rows = np.zeros(max_dets, dtype=BPBSS_ROW)

    def close(self):
        if getattr(self, "_h", None):
            lib().tlk_bpbss_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def reset(self, stream=-1):
        check(lib().tlk_bpbss_reset(self._h, stream))

    def update(self, ids, ltwh, emb, vis, conf, stream=0, keypoints=None):
        ids = np.ascontiguousarray(ids, dtype=np.int64)
        kps = None if keypoints is None else _f64(keypoints).reshape(-1, 51)
        ltwh = _f64(ltwh).reshape(-1, 4)
        emb = np.ascontiguousarray(emb, dtype=np.float32)
        vis = np.ascontiguousarray(vis, dtype=np.uint8)
        conf = _f64(conf)
        n = C.c_int(0)
        check(lib().tlk_bpbss_update(self._h, stream, ids.ctypes.data, ltwh.ctypes.data, emb.ctypes.data, vis.ctypes.data,
                                     conf.ctypes.data, None if kps is None else kps.ctypes.data, len(ids),
                                     self._rows.ctypes.data, len(self._rows), C.byref(n)))
        return self._rows[:n.value].copy()

    def update_dev(self, ids, ltwh, emb, vis, conf, counts, n_frames, rows, out_cap, out_counts, stream_ptr=None, kps=None):
        check(lib().tlk_bpbss_update_dev(self._h, ids, ltwh, emb, vis, conf, kps, counts, n_frames, rows, out_cap, out_counts,
                                         stream_ptr))

    def tracks(self, stream=0):
        cap = self.max_tracks
        ids = np.empty(cap, dtype=np.int64)
        mean, cov = np.empty((cap, 8)), np.empty((cap, 8, 8))
        feat = np.empty((cap, self.K, self.D), dtype=np.float32)
        fvis = np.empty((cap, self.K), dtype=np.uint8)
        n = C.c_int(0)
        check(lib().tlk_bpbss_get_tracks(self._h, stream, ids.ctypes.data, mean.ctypes.data, cov.ctypes.data,
                                         feat.ctypes.data, fvis.ctypes.data, cap, C.byref(n)))
        k = n.value
        return ids[:k], mean[:k], cov[:k], feat[:k], fvis[:k]


# ------------------------------------------------------------------------------------------------
# plain StrongSORT bank
# ------------------------------------------------------------------------------------------------
class SsortParams(C.Structure):
    _fields_ = [("max_dist", C.c_double), ("max_iou_dist", C.c_double), ("max_age", C.c_int32), ("max_unmatched_preds", C.c_int32),
                ("n_init", C.c_int32), ("nn_budget", C.c_int32), ("mc_lambda", C.c_double), ("ema_alpha", C.c_double),
                ("min_confidence", C.c_double), ("wrapper_mode", C.c_int32), ("img_w", C.c_int32), ("img_h", C.c_int32),
                ("dim", C.c_int32), ("max_tracks", C.c_int32), ("max_dets", C.c_int32)]


SSORT_ROW = np.dtype([("det_id", "<i8"), ("track_id", "<i8"), ("ltrb", "<f8", (4,)), ("conf", "<f8"), ("class_id", "<i4"),
                      ("tsu", "<i4")], align=True)


def _bind_ssort(L):
    if getattr(L, "_ssort_bound", False):
        return
    vp, ci = C.c_void_p, C.c_int
    L.tlk_ssort_create.argtypes = [C.POINTER(SsortParams), ci, ci, C.POINTER(vp)]
    L.tlk_ssort_destroy.argtypes = [vp]
    L.tlk_ssort_reset.argtypes = [vp, ci]
    L.tlk_ssort_update.argtypes = [vp, ci, vp, vp, ci, vp, ci, C.POINTER(ci)]
    L.tlk_ssort_update_dev.argtypes = [vp, vp, vp, vp, ci, vp, ci, vp, vp]
    L.tlk_ssort_get_tracks.argtypes = [vp, ci, vp, vp, vp, vp, vp, vp, ci, C.POINTER(ci)]
    L._ssort_bound = True


class SsortBank:
    """``n_streams`` device-resident plain-StrongSORT trackers (``tlk_ssort_*``); hyper-parameter names follow
    ``StrongSORT.__init__`` (plugins/track/strong_sort/strong_sort.py:19-32)."""

    def __init__(self, dim, max_dist=0.2, max_iou_dist=0.7, max_age=70, max_unmatched_preds=7, n_init=3, nn_budget=100,
                 mc_lambda=0.995, ema_alpha=0.9, *, min_confidence=-np.inf, wrapper_mode=False, img_w=1920, img_h=1080,
                 n_streams=1, device=0, max_tracks=256, max_dets=128, gallery_rows=4096):
        if nn_budget is None:          # the reference's unbounded gallery: every sample kept, room for `gallery_rows` per track (TlkError beyond)
            nn_budget = -int(gallery_rows)
        L = lib()
        _bind_ssort(L)
        self.params = SsortParams(max_dist, max_iou_dist, max_age, max_unmatched_preds, n_init, int(nn_budget), mc_lambda, ema_alpha,
                                  float(min_confidence), int(wrapper_mode), img_w, img_h, dim, max_tracks, max_dets)
        self.D, self.n_streams, self.max_tracks, self.max_dets = dim, n_streams, max_tracks, max_dets
        h = C.c_void_p()
        check(L.tlk_ssort_create(C.byref(self.params), n_streams, device, C.byref(h)))
        self._h = h
        self._rows = np.zeros(max_tracks, dtype=SSORT_ROW)

    def close(self):
        if getattr(self, "_h", None):
            lib().tlk_ssort_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def reset(self, stream=-1):
        check(lib().tlk_ssort_reset(self._h, stream))

    def camera_update(self, warp, stream=0, stream_ptr=None):
        """Tracker.camera_update with the ECC estimate passed in: warp = the (2,3) matrix Track.ECC returns. Call before update()."""
        L = lib()
        L.tlk_ssort_camera_update.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
        w = _f64(warp).reshape(6)
        check(L.tlk_ssort_camera_update(self._h, stream, w.ctypes.data, stream_ptr))

    def update(self, dets, feat, stream=0):
        """dets (n,7) [x1,y1,x2,y2,conf,cls,tracklab_id], feat (n,dim) -> structured rows (SSORT_ROW)."""
        dets = _f64(dets).reshape(-1, 7)
        feat = np.ascontiguousarray(feat, dtype=np.float32).reshape(len(dets), self.D)
        n = C.c_int(0)
        check(lib().tlk_ssort_update(self._h, stream, dets.ctypes.data, feat.ctypes.data, len(dets), self._rows.ctypes.data,
                                     len(self._rows), C.byref(n)))
        return self._rows[:n.value].copy()

    def update_dev(self, dets, feat, counts, n_frames, rows, out_cap, out_counts, stream_ptr=None):
        check(lib().tlk_ssort_update_dev(self._h, dets, feat, counts, n_frames, rows, out_cap, out_counts, stream_ptr))

    def tracks(self, stream=0):
        cap = self.max_tracks
        ids, st, gl = np.empty(cap, np.int64), np.empty((cap, 5), np.int64), np.empty(cap, np.int64)
        mean, cov, feat = np.empty((cap, 8)), np.empty((cap, 8, 8)), np.empty((cap, self.D), np.float32)
        n = C.c_int(0)
        check(lib().tlk_ssort_get_tracks(self._h, stream, ids.ctypes.data, mean.ctypes.data, cov.ctypes.data, feat.ctypes.data,
                                         st.ctypes.data, gl.ctypes.data, cap, C.byref(n)))
        k = n.value
        return ids[:k], mean[:k], cov[:k], feat[:k], st[:k], gl[:k]


# ------------------------------------------------------------------------------------------------
# ByteTrack bank
# ------------------------------------------------------------------------------------------------
class ByteTrackParams(C.Structure):
    _fields_ = [("track_thresh", C.c_double), ("match_thresh", C.c_double), ("frame_rate", C.c_double), ("min_confidence", C.c_double),
                ("track_buffer", C.c_int32), ("wrapper_mode", C.c_int32), ("max_tracks", C.c_int32), ("max_dets", C.c_int32)]


BYTETRACK_ROW = np.dtype([("det_id", "<i8"), ("track_id", "<i8"), ("ltrb", "<f8", (4,)), ("score", "<f8"), ("cls", "<f8")], align=True)


def _bind_bytetrack(L):
    if getattr(L, "_bt_bound", False):
        return
    vp, ci = C.c_void_p, C.c_int
    L.tlk_bytetrack_create.argtypes = [C.POINTER(ByteTrackParams), ci, ci, C.POINTER(vp)]
    L.tlk_bytetrack_destroy.argtypes = [vp]
    L.tlk_bytetrack_reset.argtypes = [vp, ci]
    L.tlk_bytetrack_reset_keep_ids.argtypes = [vp, ci]
    L.tlk_bytetrack_update.argtypes = [vp, ci, vp, ci, vp, ci, C.POINTER(ci)]
    L.tlk_bytetrack_update_dev.argtypes = [vp, vp, vp, ci, vp, ci, vp, vp]
    L.tlk_bytetrack_get_tracks.argtypes = [vp, ci, ci, vp, vp, vp, vp, ci, C.POINTER(ci)]
    L._bt_bound = True


class ByteTrackBank:
    """``n_streams`` device-resident ByteTrack trackers (``tlk_bytetrack_*``); hyper-parameter names follow
    ``BYTETracker.__init__`` (plugins/track/byte_track/byte_tracker.py:155)."""

    def __init__(self, track_thresh=0.45, match_thresh=0.8, track_buffer=25, frame_rate=30, *, min_confidence=-np.inf,
                 wrapper_mode=False, n_streams=1, device=0, max_tracks=256, max_dets=128):
        L = lib()
        _bind_bytetrack(L)
        self.params = ByteTrackParams(track_thresh, match_thresh, float(frame_rate), float(min_confidence), int(track_buffer),
                                      int(wrapper_mode), max_tracks, max_dets)
        self.n_streams, self.max_tracks, self.max_dets = n_streams, max_tracks, max_dets
        h = C.c_void_p()
        check(L.tlk_bytetrack_create(C.byref(self.params), n_streams, device, C.byref(h)))
        self._h = h
        self._rows = np.zeros(max_tracks, dtype=BYTETRACK_ROW)

    def close(self):
        if getattr(self, "_h", None):
            lib().tlk_bytetrack_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def reset(self, stream=-1, keep_ids=False):
        """keep_ids: state dropped but the id counter keeps counting, like the reference's class-level BaseTrack._count (basetrack.py:13,35-37)."""
        check((lib().tlk_bytetrack_reset_keep_ids if keep_ids else lib().tlk_bytetrack_reset)(self._h, stream))

    def update(self, dets, stream=0):
        dets = _f64(dets).reshape(-1, 7)
        n = C.c_int(0)
        check(lib().tlk_bytetrack_update(self._h, stream, dets.ctypes.data, len(dets), self._rows.ctypes.data, len(self._rows), C.byref(n)))
        return self._rows[:n.value].copy()

    def update_dev(self, dets, counts, n_frames, rows, out_cap, out_counts, stream_ptr=None):
        check(lib().tlk_bytetrack_update_dev(self._h, dets, counts, n_frames, rows, out_cap, out_counts, stream_ptr))

    def tracks(self, which=0, stream=0):
        cap = self.max_tracks
        ids, st = np.empty(cap, np.int64), np.empty((cap, 5), np.int64)
        mean, cov = np.empty((cap, 8)), np.empty((cap, 8, 8))
        n = C.c_int(0)
        check(lib().tlk_bytetrack_get_tracks(self._h, stream, which, ids.ctypes.data, mean.ctypes.data, cov.ctypes.data, st.ctypes.data, cap,
                                             C.byref(n)))
        k = n.value
        return ids[:k], mean[:k], cov[:k], st[:k]


# ------------------------------------------------------------------------------------------------
# BoT-SORT bank
# ------------------------------------------------------------------------------------------------
class BoTSORTParams(C.Structure):
    _fields_ = [("track_high_thresh", C.c_double), ("new_track_thresh", C.c_double), ("match_thresh", C.c_double),
                ("proximity_thresh", C.c_double), ("appearance_thresh", C.c_double), ("frame_rate", C.c_double), ("lambda_", C.c_double),
                ("min_confidence", C.c_double), ("track_buffer", C.c_int32), ("cmc_method", C.c_int32), ("wrapper_mode", C.c_int32),
                ("dim", C.c_int32), ("max_tracks", C.c_int32), ("max_dets", C.c_int32)]


BOTSORT_ROW = BYTETRACK_ROW
CMC_METHODS = {"none": 0, None: 0, "orb": 1, "sift": 2, "ecc": 3, "sparseOptFlow": 4, "file": 5, "files": 5}      # gmc.py:18-78


def _bind_botsort(L):
    if getattr(L, "_bo_bound", False):
        return
    vp, ci = C.c_void_p, C.c_int
    L.tlk_botsort_create.argtypes = [C.POINTER(BoTSORTParams), ci, ci, C.POINTER(vp)]
    L.tlk_botsort_destroy.argtypes = [vp]
    L.tlk_botsort_reset.argtypes = [vp, ci]
    L.tlk_botsort_reset_keep_ids.argtypes = [vp, ci]
    L.tlk_botsort_update.argtypes = [vp, ci, vp, vp, ci, vp, ci, C.POINTER(ci)]
    L.tlk_botsort_update_dev.argtypes = [vp, vp, vp, vp, ci, vp, ci, vp, vp]
    L.tlk_botsort_update_gmc.argtypes = [vp, ci, vp, vp, ci, vp, vp, ci, C.POINTER(ci)]
    L.tlk_botsort_update_dev_gmc.argtypes = [vp, vp, vp, vp, vp, ci, vp, ci, vp, vp]
    L.tlk_botsort_get_tracks.argtypes = [vp, ci, ci, vp, vp, vp, vp, vp, ci, C.POINTER(ci)]
    L._bo_bound = True


class BoTSORTBank:
    """``n_streams`` device-resident BoT-SORT trackers (``tlk_botsort_*``); hyper-parameter names follow ``BoTSORT.__init__``
    (plugins/track/bot_sort/bot_sort.py:236-249). With a ``cmc_method`` other than "none" every update takes the frame's (2,3) warp
    (``update(..., warp=H)`` / ``update_dev(..., warps=ptr)``): the bank applies it on the device (STrack.multi_gmc), the estimate comes
    from ``tracklab_amd.cmc`` (tlk_cmc_*) or any other source."""

    def __init__(self, dim, track_high_thresh=0.45, new_track_thresh=0.6, track_buffer=30, match_thresh=0.8, proximity_thresh=0.5,
                 appearance_thresh=0.25, cmc_method="none", frame_rate=30, lambda_=0.985, *, min_confidence=-np.inf, wrapper_mode=False,
                 n_streams=1, device=0, max_tracks=256, max_dets=128):
        L = lib()
        _bind_botsort(L)
        if cmc_method not in CMC_METHODS:
            raise ValueError(f"Unknown CMC method: {cmc_method}")                  # gmc.py:80
        self.params = BoTSORTParams(track_high_thresh, new_track_thresh, match_thresh, proximity_thresh, appearance_thresh, float(frame_rate),
                                    lambda_, float(min_confidence), int(track_buffer), CMC_METHODS[cmc_method], int(wrapper_mode), int(dim),
                                    max_tracks, max_dets)
        self.n_streams, self.max_tracks, self.max_dets, self.dim = n_streams, max_tracks, max_dets, int(dim)
        h = C.c_void_p()
        check(L.tlk_botsort_create(C.byref(self.params), n_streams, device, C.byref(h)))
        self._h = h
        self._rows = np.zeros(max_tracks, dtype=BOTSORT_ROW)

    def close(self):
        if getattr(self, "_h", None):
            lib().tlk_botsort_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def reset(self, stream=-1, keep_ids=False):
        """keep_ids: state dropped but the id counter keeps counting, like the reference's class-level BaseTrack._count."""
        check((lib().tlk_botsort_reset_keep_ids if keep_ids else lib().tlk_botsort_reset)(self._h, stream))

    def update(self, dets, feats, stream=0, warp=None):
        """warp: the frame's (2,3) camera-motion matrix (GMC.apply's return value, bot_sort.py:341) or None for the identity."""
        dets = _f64(dets).reshape(-1, 7)
        feats = np.ascontiguousarray(feats, dtype=np.float32).reshape(len(dets), self.dim)
        n = C.c_int(0)
        w = None if warp is None else _f64(warp).reshape(6)
        check(lib().tlk_botsort_update_gmc(self._h, stream, dets.ctypes.data, feats.ctypes.data, len(dets), None if w is None else w.ctypes.data,
                                           self._rows.ctypes.data, len(self._rows), C.byref(n)))
        return self._rows[:n.value].copy()

    def update_dev(self, dets, feats, counts, n_frames, rows, out_cap, out_counts, stream_ptr=None, warps=None):
        """warps: device pointer to (S, n_frames, 6) float64 warps, or None (identity)."""
        check(lib().tlk_botsort_update_dev_gmc(self._h, dets, feats, counts, warps, n_frames, rows, out_cap, out_counts, stream_ptr))

    def tracks(self, which=0, stream=0):
        cap = self.max_tracks
        ids, st = np.empty(cap, np.int64), np.empty((cap, 5), np.int64)
        mean, cov, feat = np.empty((cap, 8)), np.empty((cap, 8, 8)), np.empty((cap, self.dim), np.float32)
        n = C.c_int(0)
        check(lib().tlk_botsort_get_tracks(self._h, stream, which, ids.ctypes.data, mean.ctypes.data, cov.ctypes.data, st.ctypes.data,
                                           feat.ctypes.data, cap, C.byref(n)))
        k = n.value
        return ids[:k], mean[:k], cov[:k], st[:k], feat[:k]


# ------------------------------------------------------------------------------------------------
# Deep-OC-SORT bank
# ------------------------------------------------------------------------------------------------
class DeepOCSortParams(C.Structure):
    _fields_ = [("det_thresh", C.c_double), ("iou_threshold", C.c_double), ("inertia", C.c_double), ("w_association_emb", C.c_double),
                ("alpha_fixed_emb", C.c_double), ("aw_param", C.c_double), ("min_confidence", C.c_double),
                ("max_age", C.c_int32), ("min_hits", C.c_int32), ("delta_t", C.c_int32), ("asso_func", C.c_int32),
                ("embedding_off", C.c_int32), ("cmc_off", C.c_int32), ("aw_off", C.c_int32), ("new_kf_off", C.c_int32),
                ("wrapper_mode", C.c_int32), ("dim", C.c_int32), ("max_tracks", C.c_int32), ("max_dets", C.c_int32)]


# rows of tlk_deepocsort_update(_dev) viewed with the field names of the other banks (ocsort.py:527-529)
DEEPOCSORT_ROW = np.dtype([("ltrb", "<f8", (4,)), ("track_id", "<f8"), ("cls", "<f8"), ("conf", "<f8"), ("det_id", "<f8")])


def _bind_deepocsort(L):
    if getattr(L, "_doc_bound", False):
        return
    vp, ci = C.c_void_p, C.c_int
    L.tlk_deepocsort_create.argtypes = [C.POINTER(DeepOCSortParams), ci, ci, C.POINTER(vp)]
    L.tlk_deepocsort_destroy.argtypes = [vp]
    L.tlk_deepocsort_reset.argtypes = [vp, ci]
    L.tlk_deepocsort_update.argtypes = [vp, ci, vp, vp, ci, vp, ci, C.POINTER(ci)]
    L.tlk_deepocsort_update_dev.argtypes = [vp, vp, vp, vp, ci, vp, ci, vp, vp]
    L.tlk_deepocsort_get_tracks.argtypes = [vp, ci, vp, vp, vp, vp, vp, vp, vp, ci, C.POINTER(ci)]
    L._doc_bound = True


class DeepOCSortBank:
    """``n_streams`` device-resident Deep-OC-SORT trackers (``tlk_deepocsort_*``); hyper-parameter names follow ``OCSort.__init__``
    (plugins/track/deep_oc_sort/ocsort.py:352-371). cmc_off must be true, embedding_off / new_kf_off false (TlkError otherwise)."""

    def __init__(self, dim, det_thresh=0.0, max_age=30, min_hits=3, iou_threshold=0.3, delta_t=3, asso_func="iou", inertia=0.2,
                 w_association_emb=0.75, alpha_fixed_emb=0.95, aw_param=0.5, embedding_off=False, cmc_off=False, aw_off=False,
                 new_kf_off=False, *, min_confidence=-np.inf, wrapper_mode=False, n_streams=1, device=0, max_tracks=256, max_dets=128):
        L = lib()
        _bind_deepocsort(L)
        self.params = DeepOCSortParams(det_thresh, iou_threshold, inertia, w_association_emb, alpha_fixed_emb, aw_param, float(min_confidence),
                                       int(max_age), int(min_hits), int(delta_t), ASSO[asso_func], int(embedding_off), int(cmc_off),
                                       int(aw_off), int(new_kf_off), int(wrapper_mode), int(dim), max_tracks, max_dets)
        self.n_streams, self.max_tracks, self.max_dets, self.dim = n_streams, max_tracks, max_dets, int(dim)
        h = C.c_void_p()
        check(L.tlk_deepocsort_create(C.byref(self.params), n_streams, device, C.byref(h)))
        self._h = h
        self._out = np.zeros((max_tracks + max_dets, 8))

    def close(self):
        if getattr(self, "_h", None):
            lib().tlk_deepocsort_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def reset(self, stream=-1):
        check(lib().tlk_deepocsort_reset(self._h, stream))

    def affine_correction(self, warp, stream=0, stream_ptr=None):
        """apply_affine_correction of every tracker with the estimate passed in: warp = the (2,3) affine compute_affine returns.
        Call before update() (the reference applies it ahead of predict, ocsort.py:425-428)."""
        L = lib()
        L.tlk_deepocsort_affine_correction.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
        w = _f64(warp).reshape(6)
        check(L.tlk_deepocsort_affine_correction(self._h, stream, w.ctypes.data, stream_ptr))

    def update(self, dets, embs, stream=0):
        """dets (n,7), embs (n,dim) float32 -> rows (m,8) [x1,y1,x2,y2,track_id,cls,conf,tracklab_id]."""
        dets = _f64(dets).reshape(-1, 7)
        embs = np.ascontiguousarray(embs, dtype=np.float32).reshape(len(dets), self.dim)
        n = C.c_int(0)
        check(lib().tlk_deepocsort_update(self._h, stream, dets.ctypes.data, embs.ctypes.data, len(dets), self._out.ctypes.data, len(self._out),
                                          C.byref(n)))
        return self._out[:n.value].copy()

    def update_dev(self, dets, embs, counts, n_frames, out, out_cap, out_counts, stream_ptr=None):
        check(lib().tlk_deepocsort_update_dev(self._h, dets, embs, counts, n_frames, out, out_cap, out_counts, stream_ptr))

    def tracks(self, stream=0):
        cap = self.max_tracks
        ids, st = np.empty(cap, np.int64), np.empty((cap, 6), np.int64)
        x, P, emb = np.empty((cap, 8)), np.empty((cap, 8, 8)), np.empty((cap, self.dim), np.float32)
        vel, last = np.empty((cap, 2)), np.empty((cap, 5))
        n = C.c_int(0)
        check(lib().tlk_deepocsort_get_tracks(self._h, stream, ids.ctypes.data, x.ctypes.data, P.ctypes.data, emb.ctypes.data, st.ctypes.data,
                                              vel.ctypes.data, last.ctypes.data, cap, C.byref(n)))
        k = n.value
        return ids[:k], x[:k], P[:k], emb[:k], st[:k], vel[:k], last[:k]


def partdist(q, qvis, g, gvis):
    """q (T,K,D) f32, qvis (T,K) u8, g (N,K,D) f32, gvis (N,K) u8 cuda tensors -> (T,N) f64 cuda tensor."""
    import torch
    L = lib()
    _bind_bpbss(L)
    T, K, D = q.shape
    N = g.shape[0]
    assert q.is_cuda and q.dtype == torch.float32 and q.is_contiguous() and g.is_contiguous()
    assert qvis.dtype == torch.uint8 and gvis.dtype == torch.uint8
    out = torch.empty((T, N), dtype=torch.float64, device=q.device)
    check(L.tlk_partdist_f32(q.data_ptr(), qvis.data_ptr(), T, g.data_ptr(), gvis.data_ptr(), N, K, D, out.data_ptr(),
                             current_stream_ptr()))
    return out


_GEMM_OK = None          # None: untried, True / False afterwards


def gemm_bias_act(x2d, weight2d, bias, act="relu", residual2d=None):
    """out (M, N) = act(x2d (M, K) @ weight2d (N, K)^T + bias (+ residual2d (M, N))) in ONE hipBLASLt call (tlk_gemm_bias_act).
    Returns None when that route is unavailable (callers then use GEMM + tlk_bias_act_nhwc)."""
    global _GEMM_OK
    import torch
    if _GEMM_OK is False:
        return None
    L = lib()
    if not getattr(L, "_gemm_bound", False):
        L.tlk_gemm_bias_act.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_longlong, C.c_int, C.c_int,
                                        C.c_int, C.c_int, C.c_void_p]
        L._gemm_bound = True
    M, K = x2d.shape
    N = weight2d.shape[0]
    out = torch.empty((M, N), dtype=x2d.dtype, device=x2d.device)
    rc = L.tlk_gemm_bias_act(x2d.data_ptr(), weight2d.data_ptr(), bias.data_ptr(), residual2d.data_ptr() if residual2d is not None else None,
                             out.data_ptr(), M, N, K, ACT[act], _dtype_code(x2d.dtype), current_stream_ptr())
    if rc == -5:            # TLK_EUNSUPPORTED
        if _GEMM_OK is None:
            _GEMM_OK = False
        return None
    check(rc)
    _GEMM_OK = True
    return out


# ------------------------------------------------------------------------------------------------
# stateless Kalman steps and motion costs (include/tlk.h: tlk_kf7_*, tlk_kf8_*, tlk_iou_ltwh_cost_f64, tlk_oks_cost_f64)
# All take/return float64 cuda tensors; the in-place ones return their (modified) arguments.
# ------------------------------------------------------------------------------------------------
def _bind_kf(L):
    if getattr(L, "_kf_bound", False):
        return
    vp, ci = C.c_void_p, C.c_int
    L.tlk_kf7_predict_f64.argtypes = [vp, vp, ci, vp]
    L.tlk_kf7_update_f64.argtypes = [vp, vp, vp, ci, vp]
    L.tlk_kf8_initiate_f64.argtypes = [vp, vp, vp, ci, vp]
    L.tlk_kf8_predict_f64.argtypes = [vp, vp, ci, vp]
    L.tlk_kf8_project_f64.argtypes = [vp, vp, vp, vp, vp, ci, vp]
    L.tlk_kf8_update_f64.argtypes = [vp, vp, vp, vp, ci, vp]
    L.tlk_kf8_gate_f64.argtypes = [vp, vp, ci, vp, ci, ci, vp, vp]
    L.tlk_iou_ltwh_cost_f64.argtypes = [vp, ci, vp, ci, vp, vp]
    L.tlk_oks_cost_f64.argtypes = [vp, ci, vp, ci, vp, vp]
    L._kf_bound = True


def _f64c(t, *shape_tail):
    import torch
    assert t.is_cuda and t.dtype == torch.float64 and t.is_contiguous(), "float64 contiguous cuda tensor expected"
    assert tuple(t.shape[1:]) == tuple(shape_tail), f"shape (n,{shape_tail}) expected, got {tuple(t.shape)}"
    return t


def kf7_predict_(x, P):
    L = lib(); _bind_kf(L)
    n = _f64c(x, 7).shape[0]; _f64c(P, 7, 7)
    check(L.tlk_kf7_predict_f64(x.data_ptr(), P.data_ptr(), n, current_stream_ptr()))
    return x, P


def kf7_update_(x, P, z):
    L = lib(); _bind_kf(L)
    n = _f64c(x, 7).shape[0]; _f64c(P, 7, 7); _f64c(z, 4)
    check(L.tlk_kf7_update_f64(x.data_ptr(), P.data_ptr(), z.data_ptr(), n, current_stream_ptr()))
    return x, P


def kf8_initiate(meas):
    import torch
    L = lib(); _bind_kf(L)
    n = _f64c(meas, 4).shape[0]
    mean = torch.empty((n, 8), dtype=torch.float64, device=meas.device)
    cov = torch.empty((n, 8, 8), dtype=torch.float64, device=meas.device)
    check(L.tlk_kf8_initiate_f64(meas.data_ptr(), mean.data_ptr(), cov.data_ptr(), n, current_stream_ptr()))
    return mean, cov


def kf8_predict_(mean, cov):
    L = lib(); _bind_kf(L)
    n = _f64c(mean, 8).shape[0]; _f64c(cov, 8, 8)
    check(L.tlk_kf8_predict_f64(mean.data_ptr(), cov.data_ptr(), n, current_stream_ptr()))
    return mean, cov


def kf8_project(mean, cov, conf=None):
    import torch
    L = lib(); _bind_kf(L)
    n = _f64c(mean, 8).shape[0]; _f64c(cov, 8, 8)
    if conf is not None:
        _f64c(conf)
    pm = torch.empty((n, 4), dtype=torch.float64, device=mean.device)
    pc = torch.empty((n, 4, 4), dtype=torch.float64, device=mean.device)
    check(L.tlk_kf8_project_f64(mean.data_ptr(), cov.data_ptr(), conf.data_ptr() if conf is not None else None, pm.data_ptr(),
                                pc.data_ptr(), n, current_stream_ptr()))
    return pm, pc


def kf8_update_(mean, cov, meas, conf=None):
    L = lib(); _bind_kf(L)
    n = _f64c(mean, 8).shape[0]; _f64c(cov, 8, 8); _f64c(meas, 4)
    if conf is not None:
        _f64c(conf)
    check(L.tlk_kf8_update_f64(mean.data_ptr(), cov.data_ptr(), meas.data_ptr(), conf.data_ptr() if conf is not None else None, n,
                               current_stream_ptr()))
    return mean, cov


def kf8_gate(mean, cov, meas, only_position=False):
    import torch
    L = lib(); _bind_kf(L)
    T = _f64c(mean, 8).shape[0]; _f64c(cov, 8, 8)
    N = _f64c(meas, 4).shape[0]
    out = torch.empty((T, N), dtype=torch.float64, device=mean.device)
    check(L.tlk_kf8_gate_f64(mean.data_ptr(), cov.data_ptr(), T, meas.data_ptr(), N, int(bool(only_position)), out.data_ptr(),
                             current_stream_ptr()))
    return out


def iou_ltwh_cost(tracks_ltwh, dets_ltwh):
    import torch
    L = lib(); _bind_kf(L)
    T = _f64c(tracks_ltwh, 4).shape[0]
    N = _f64c(dets_ltwh, 4).shape[0]
    out = torch.empty((T, N), dtype=torch.float64, device=tracks_ltwh.device)
    check(L.tlk_iou_ltwh_cost_f64(tracks_ltwh.data_ptr(), T, dets_ltwh.data_ptr(), N, out.data_ptr(), current_stream_ptr()))
    return out


def oks_cost(track_kps, det_kps):
    import torch
    L = lib(); _bind_kf(L)
    T = _f64c(track_kps, 17, 3).shape[0]
    N = _f64c(det_kps, 17, 3).shape[0]
    out = torch.empty((T, N), dtype=torch.float64, device=track_kps.device)
    check(L.tlk_oks_cost_f64(track_kps.data_ptr(), T, det_kps.data_ptr(), N, out.data_ptr(), current_stream_ptr()))
    return out


# ------------------------------------------------------------------------------------------------
# fused conv epilogue (bias + activation (+ residual)) for channels-last backbones
# ------------------------------------------------------------------------------------------------
ACT = {None: 0, "none": 0, "relu": 1, "silu": 2}
ACT_RES_AFTER = 0x100          # TLK_ACT_RES_AFTER: y = act(conv + bias) + residual instead of act(conv + bias + residual)


def bias_act_(x, bias, act="relu", residual=None):
    """In place: x = act(x + bias[c] (+ residual)) for a channels-last (N,C,H,W) fp16/bf16 cuda tensor."""
    import torch
    L = lib()
    if not getattr(L, "_epi_bound", False):
        L.tlk_bias_act_nhwc.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_longlong, C.c_int, C.c_int, C.c_int, C.c_void_p]
        L._epi_bound = True
    N, Cc, H, W = x.shape
    assert x.is_contiguous(memory_format=torch.channels_last) and bias.dtype == x.dtype and bias.is_contiguous()
    if residual is not None:
        assert residual.shape == x.shape and residual.dtype == x.dtype and residual.is_contiguous(memory_format=torch.channels_last)
    check(L.tlk_bias_act_nhwc(x.data_ptr(), bias.data_ptr(), residual.data_ptr() if residual is not None else None,
                              N * H * W, Cc, ACT[act], _dtype_code(x.dtype), current_stream_ptr()))
    return x


def conv2d_nhwc_f32(x, weight, bias=None, act=None, residual=None, stride=1, pad=None, out=None, residual_after_act=False):
    """fp32 convolution + bias + residual + activation in ONE hand-written MFMA kernel (tlk_conv2d_nhwc_f32).
    x: (N, Cin, H, W) float32 cuda tensor in channels_last memory (or a channel slice of one); weight: (Cout, Cin, KH, KW) channels_last;
    residual / out: (N, Cout, Ho, Wo) channels_last (or channel slices).  Returns out (allocated channels_last when None)."""
    import torch
    L = lib()
    if not getattr(L, "_conv_bound", False):
        L.tlk_conv2d_nhwc_f32.argtypes = [C.c_void_p] * 5 + [C.c_int] * 13 + [C.c_void_p]
        L.tlk_conv2d_set_config.argtypes = [C.c_int]
        L._conv_bound = True
    N, Cin, H, W = x.shape
    Cout, Cw, KH, KW = weight.shape
    if pad is None:
        pad = (KH - 1) // 2
    assert x.dtype == torch.float32 and weight.dtype == torch.float32 and Cw == Cin
    Ho, Wo = (H + 2 * pad - KH) // stride + 1, (W + 2 * pad - KW) // stride + 1

    pix = _pix16               # pixel stride in elements (size-1 dimensions carry arbitrary strides in torch: ADVICE r04)
    wk = weight if weight.is_contiguous(memory_format=torch.channels_last) or (KH == 1 and KW == 1 and weight.is_contiguous()) \
        else weight.contiguous(memory_format=torch.channels_last)
    if out is None:
        out = torch.empty((N, Cout, Ho, Wo), dtype=torch.float32, device=x.device, memory_format=torch.channels_last)
    check(L.tlk_conv2d_nhwc_f32(x.data_ptr(), wk.data_ptr(), bias.data_ptr() if bias is not None else None,
                                residual.data_ptr() if residual is not None else None, out.data_ptr(),
                                N, H, W, Cin, Cout, KH, KW, stride, pad, ACT[act] | (ACT_RES_AFTER if residual_after_act else 0),
                                pix(x, Cin, H, W), pix(out, Cout, Ho, Wo), pix(residual, Cout, Ho, Wo) if residual is not None else 0,
                                current_stream_ptr()))
    return out


# tlk_conv2d_nhwc_f32 configurations 21..: the direct-to-LDS kernels of csrc/tlk_conv16x.hip on fp32 tensors, as rocprofv3 names them --
# conv16x_kernel<WGM, WGN, TM, TN, MODE_F32 = 2, NST, RESPF, PATCH> (activation and residual are run-time switches there)
CONV_F32_X_TEMPLATES = {21: "2, 2, 2, 2, 2, 1, true, false, 128, 1", 22: "2, 2, 2, 2, 2, 1, false, false, 128, 1", 23: "4, 1, 2, 2, 2, 1, false, false, 128, 1",
                        24: "2, 2, 2, 2, 2, 2, true, false, 128, 1", 25: "2, 2, 1, 2, 2, 1, true, false, 128, 1", 26: "4, 1, 2, 2, 2, 1, true, false, 128, 1",
                        27: "4, 1, 2, 1, 2, 1, true, false, 128, 1", 28: "4, 1, 2, 1, 2, 2, true, false, 128, 1", 29: "4, 1, 4, 1, 2, 1, true, false, 128, 1",
                        30: "4, 1, 2, 1, 2, 1, true, true, 128, 1", 31: "4, 1, 2, 1, 2, 1, false, true, 128, 1", 32: "4, 1, 1, 1, 2, 1, true, true, 128, 1",
                        33: "2, 1, 2, 1, 2, 1, true, true, 128, 1", 34: "4, 1, 1, 2, 2, 1, true, true, 128, 2", 35: "4, 1, 2, 2, 2, 1, true, true, 128, 2",
                        36: "2, 1, 2, 2, 2, 1, true, true, 128, 2", 37: "4, 1, 1, 2, 2, 1, false, true, 128, 2",
                        38: "2, 2, 2, 2, 2, 2, false, false, 128, 1", 39: "4, 2, 2, 2, 2, 2, false, false, 128, 1"}


def conv_f32_config_template(cfg: int) -> str:
    return CONV_F32_X_TEMPLATES.get(int(cfg), f"configuration {int(cfg)}")


def _pix16(t, c, h, w):
    """pixel stride (elements) of an NHWC tensor or channel slice of one, logical shape (N, C, H, W)"""
    sn, sc, sh, sw = t.stride()
    if t.shape[0] == 0:
        return c
    # (strides of size-1 dimensions are arbitrary in torch: the pixel stride is read from the first spatial dimension that has extent)
    pix = sw if w > 1 else (sh if h > 1 else (sn if t.shape[0] > 1 else c))
    assert (sc == 1 or c == 1) and (w == 1 or h == 1 or sh == w * sw) and (t.shape[0] == 1 or sn == h * w * pix), \
        "tensor must be channels_last (or a channel slice of one)"
    return pix


def yolox_head(cls_feats, reg_feats, weights, biases, num_classes=1, out=None):
    """YOLOX's decoupled head outputs for all pyramid levels in ONE launch (``tlk_yolox_head_nhwc``, r06).
    cls_feats / reg_feats: per level (B, C, H, W) cuda tensors in channels_last memory (float32 or float16: the branch outputs); weights: per level
    (5 + num_classes, C) float32 [reg 0..3 | obj | cls...]; biases (5 + num_classes,) float32.  Returns (B, A, 5 + num_classes) float32:
    [reg raw, sigmoid(obj), sigmoid(cls)], levels concatenated along the anchors -- what ``yolox_decode_nms`` consumes."""
    import torch
    L = lib()
    if not getattr(L, "_heads_bound", False):
        _bind_heads(L)
    nl = len(cls_feats)
    x0 = cls_feats[0]
    B, Cc = x0.shape[0], x0.shape[1]
    hw = [int(f.shape[2] * f.shape[3]) for f in cls_feats]
    A = sum(hw)
    NO = 5 + num_classes
    if out is None:
        out = torch.empty((B, A, NO), dtype=torch.float32, device=x0.device)
    assert out.shape == (B, A, NO) and out.dtype == torch.float32 and out.is_contiguous()
    for cf, rf, w, b in zip(cls_feats, reg_feats, weights, biases):
        assert cf.dtype == x0.dtype and rf.dtype == x0.dtype and cf.shape == rf.shape and cf.shape[:2] == (B, Cc)
        assert w.dtype == torch.float32 and w.shape == (NO, Cc) and w.is_contiguous() and b.dtype == torch.float32 and b.shape == (NO,)
    vp = C.c_void_p * nl
    ip = C.c_int * nl
    check(L.tlk_yolox_head_nhwc(vp(*[f.data_ptr() for f in cls_feats]), vp(*[f.data_ptr() for f in reg_feats]), ip(*hw),
                                ip(*[_pix16(f, Cc, f.shape[2], f.shape[3]) for f in cls_feats]), ip(*[_pix16(f, Cc, f.shape[2], f.shape[3]) for f in reg_feats]),
                                vp(*[w.data_ptr() for w in weights]), vp(*[b.data_ptr() for b in biases]), nl, B, Cc, num_classes,
                                _dtype_code(x0.dtype), out.data_ptr(), current_stream_ptr()))
    return out


def reid_part_head(feat, weight, bias, vis_threshold, counts=None, slot_base=None, max_dets=0, out_emb=None, out_vis=None, flag=None):
    """The part-based ReID head in ONE launch (``tlk_reid_part_head``, r06): feat (N, D, h, w) cuda tensor in channels_last memory (float32 /
    float16), weight (K, D) float32, bias (K,) float32 -> emb (rows, K, D) float32, vis (rows, K) uint8.  counts (frames,) int32 + max_dets:
    rows = frames * max_dets in the tracker's (frame, slot) layout, padding rows zero-filled; slot_base (frames,) int32: the rows are read from
    the DENSE batch (``crop_slot_bases``).  flag: 1-element bool / uint8 cuda tensor, set when a live embedding is not finite.
    vis_threshold: the model's threshold already divided by the part count."""
    import torch
    L = lib()
    if not getattr(L, "_heads_bound", False):
        _bind_heads(L)
    N, D, h, w = feat.shape
    K = weight.shape[0]
    assert feat.is_cuda and feat.dtype in (torch.float32, torch.float16)
    assert weight.dtype == torch.float32 and weight.shape == (K, D) and weight.is_contiguous() and bias.dtype == torch.float32 and bias.shape == (K,)
    rows = N if counts is None else counts.numel() * max_dets
    if counts is not None:
        assert counts.dtype == torch.int32 and counts.is_contiguous() and max_dets > 0 and (slot_base is not None or rows <= N)
    if slot_base is not None:
        assert slot_base.dtype == torch.int32 and slot_base.numel() == counts.numel() and slot_base.is_contiguous()
    if out_emb is None:
        out_emb = torch.empty((rows, K, D), dtype=torch.float32, device=feat.device)
    if out_vis is None:
        out_vis = torch.empty((rows, K), dtype=torch.uint8, device=feat.device)
    assert out_emb.numel() == rows * K * D and out_emb.dtype == torch.float32 and out_emb.is_contiguous()
    assert out_vis.numel() == rows * K and out_vis.dtype in (torch.uint8, torch.bool) and out_vis.is_contiguous()
    assert flag is None or (flag.numel() == 1 and flag.dtype in (torch.uint8, torch.bool))
    check(L.tlk_reid_part_head(feat.data_ptr(), _pix16(feat, D, h, w), _dtype_code(feat.dtype), h * w, D, K, weight.data_ptr(), bias.data_ptr(),
                               counts.data_ptr() if counts is not None else None, slot_base.data_ptr() if slot_base is not None else None,
                               rows, max_dets, float(vis_threshold), out_emb.data_ptr(), out_vis.data_ptr(),
                               flag.data_ptr() if flag is not None else None, current_stream_ptr()))
    return out_emb, out_vis


def _bind_heads(L):
    L.tlk_yolox_head_nhwc.argtypes = [C.c_void_p] * 7 + [C.c_int] * 5 + [C.c_void_p, C.c_void_p]
    L.tlk_reid_part_head.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                     C.c_int, C.c_int, C.c_float, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    L._heads_bound = True


def dwconv_set_config(cfg):
    """probes: lane shape of tlk_dwconv2d_nhwc (0 = by element type, 1..4 = (columns per lane, rows ahead) = (1, 1), (2, 1), (1, 2), (2, 2))"""
    L = lib()
    L.tlk_dwconv_set_config.argtypes = [C.c_int]
    check(L.tlk_dwconv_set_config(int(cfg)))


def dwconv2d_nhwc(x, weight_kkc, bias32=None, act=None, out=None):
    """Depthwise k x k convolution (stride 1, pad k // 2) + bias + activation in ONE hand-written kernel (tlk_dwconv2d_nhwc).
    x: (N, C, H, W) float32 / float16 cuda tensor in channels_last memory (or a channel slice of one); weight_kkc: (k, k, C) contiguous, the
    dtype of x (torch's (C, 1, k, k) depthwise weight as `w.permute(2, 3, 0, 1).reshape(k, k, C)`); bias32: (C,) float32 or None."""
    import torch
    L = lib()
    if not getattr(L, "_dwconv_bound", False):
        L.tlk_dwconv2d_nhwc.argtypes = [C.c_void_p] * 4 + [C.c_int] * 9 + [C.c_void_p]
        L._dwconv_bound = True
    N, Cc, H, W = x.shape
    k = weight_kkc.shape[0]
    assert weight_kkc.shape == (k, k, Cc) and weight_kkc.is_contiguous() and weight_kkc.dtype == x.dtype
    assert x.dtype in (torch.float32, torch.float16) and (bias32 is None or (bias32.dtype == torch.float32 and bias32.is_contiguous()))
    if out is None:
        out = torch.empty((N, Cc, H, W), dtype=x.dtype, device=x.device, memory_format=torch.channels_last)
    check(L.tlk_dwconv2d_nhwc(x.data_ptr(), weight_kkc.data_ptr(), bias32.data_ptr() if bias32 is not None else None, out.data_ptr(),
                              N, H, W, Cc, k, ACT[act], _dtype_code(x.dtype), _pix16(x, Cc, H, W), _pix16(out, Cc, H, W), current_stream_ptr()))
    return out


def spp_maxpool_nhwc(x, out=None):
    """[x | maxpool5(x) | maxpool9(x) | maxpool13(x)] (stride 1, same size) along the channels in ONE pass (tlk_spp_maxpool_nhwc).
    x: (N, C, H, W) float32 / float16 cuda tensor in channels_last memory (or a channel slice of one) -> (N, 4C, H, W) channels_last."""
    import torch
    L = lib()
    if not getattr(L, "_spp_bound", False):
        L.tlk_spp_maxpool_nhwc.argtypes = [C.c_void_p] * 2 + [C.c_int] * 7 + [C.c_void_p]
        L._spp_bound = True
    N, Cc, H, W = x.shape
    assert x.dtype in (torch.float32, torch.float16)
    if out is None:
        out = torch.empty((N, 4 * Cc, H, W), dtype=x.dtype, device=x.device, memory_format=torch.channels_last)
    check(L.tlk_spp_maxpool_nhwc(x.data_ptr(), out.data_ptr(), N, H, W, Cc, _dtype_code(x.dtype), _pix16(x, Cc, H, W), _pix16(out, 4 * Cc, H, W),
                                 current_stream_ptr()))
    return out


def maxpool2d_nhwc(x, k=3, stride=2, pad=1, out=None):
    """k x k max pooling (stride, -inf padding) of a channels_last (N, C, H, W) float32 / float16 tensor in ONE hand-written pass
    (tlk_maxpool2d_nhwc); C a multiple of 16 bytes of elements."""
    import torch
    L = lib()
    if not getattr(L, "_maxpool_bound", False):
        L.tlk_maxpool2d_nhwc.argtypes = [C.c_void_p] * 2 + [C.c_int] * 10 + [C.c_void_p]
        L._maxpool_bound = True
    N, Cc, H, W = x.shape
    assert x.dtype in (torch.float32, torch.float16)
    Ho, Wo = (H + 2 * pad - k) // stride + 1, (W + 2 * pad - k) // stride + 1
    if out is None:
        out = torch.empty((N, Cc, Ho, Wo), dtype=x.dtype, device=x.device, memory_format=torch.channels_last)
    check(L.tlk_maxpool2d_nhwc(x.data_ptr(), out.data_ptr(), N, H, W, Cc, k, stride, pad, _dtype_code(x.dtype), _pix16(x, Cc, H, W),
                               _pix16(out, Cc, Ho, Wo), current_stream_ptr()))
    return out


def conv_stem16_pack(weight):
    """(Cout, 3, KH, KW) float16 stem weight -> the fragment-order buffer tlk_conv_stem16_nhwc multiplies from (tlk_conv_stem16_pack).  The caller
    caches the result per weight version."""
    import torch
    L = lib()
    if not getattr(L, "_stem16_bound", False):
        L.tlk_conv_stem16_packed_halfs.restype = C.c_longlong
        L.tlk_conv_stem16_packed_halfs.argtypes = [C.c_int] * 4
        L.tlk_conv_stem16_pack.argtypes = [C.c_void_p, C.c_void_p] + [C.c_int] * 4 + [C.c_void_p]
        L.tlk_conv_stem16_nhwc.argtypes = [C.c_void_p] * 4 + [C.c_int] * 12 + [C.c_void_p]
        L._stem16_bound = True
    cout, cin, kh, kw = weight.shape
    assert cin == 3 and weight.dtype == torch.float16 and weight.is_cuda
    nh = L.tlk_conv_stem16_packed_halfs(cout, kh, kw, 2)
    if nh < 0:
        check(int(nh))
    w = weight.detach().permute(0, 2, 3, 1).contiguous()                # (Cout, KH, KW, 3)
    packed = torch.empty(int(nh), dtype=torch.float16, device=weight.device)
    check(L.tlk_conv_stem16_pack(w.data_ptr(), packed.data_ptr(), cout, kh, kw, 2, current_stream_ptr()))
    return packed


def conv_stem16(x, packed_weight, cout, k, bias32=None, act=None, pool=False, out=None):
    """RGB stem in f16 (tlk_conv_stem16_nhwc): k x k / stride 2 / pad k // 2 convolution of a channels_last (N, 3, H, W) float16 tensor + bias +
    activation, with ResNet's 3 x 3 / stride 2 / pad 1 max-pool fused behind it when `pool`.  Returns channels_last float16."""
    import torch
    L = lib()
    N, Cc, H, W = x.shape
    assert Cc == 3 and x.dtype == torch.float16 and x.is_cuda
    pad = k // 2
    Ho, Wo = (H + 2 * pad - k) // 2 + 1, (W + 2 * pad - k) // 2 + 1
    oh, ow = ((Ho - 1) // 2 + 1, (Wo - 1) // 2 + 1) if pool else (Ho, Wo)
    if out is None:
        out = torch.empty((N, cout, oh, ow), dtype=torch.float16, device=x.device, memory_format=torch.channels_last)
    check(L.tlk_conv_stem16_nhwc(x.data_ptr(), packed_weight.data_ptr(), bias32.data_ptr() if bias32 is not None else None, out.data_ptr(), N, H, W, cout,
                                 k, k, 2, pad, ACT[act], 1 if pool else 0, _pix16(x, 3, H, W), _pix16(out, cout, oh, ow), current_stream_ptr()))
    return out


def _bind_conv16(L):
    if not getattr(L, "_conv16_bound", False):
        L.tlk_conv2d_nhwc_16.argtypes = [C.c_void_p] * 10 + [C.c_int] * 13 + [C.c_void_p]
        L.tlk_split_f32_planes.argtypes = [C.c_void_p, C.c_longlong, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
        L.tlk_merge_planes_f32.argtypes = [C.c_void_p, C.c_void_p, C.c_longlong, C.c_void_p, C.c_void_p]
        L.tlk_conv2d_nhwc_16s.argtypes = [C.c_void_p] * 10 + [C.c_int] * 13 + [C.c_void_p] * 4
        L.tlk_split_f32_planes_s.argtypes = [C.c_void_p, C.c_longlong, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_longlong, C.c_void_p]
        L.tlk_merge_planes_f32_s.argtypes = [C.c_void_p, C.c_void_p, C.c_longlong, C.c_void_p, C.c_void_p, C.c_void_p]
        L.tlk_split_scale_update.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
        L.tlk_split_fuse_sum.argtypes = [C.c_int] + [C.c_void_p] * 6 + [C.c_int] * 5 + [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p]
        L.tlk_fuse_sum_f32.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p] + [C.c_int] * 5 + [C.c_void_p, C.c_int, C.c_int, C.c_void_p]
        L.tlk_fuse_sum_f16.argtypes = L.tlk_fuse_sum_f32.argtypes
        L._conv16_bound = True


def conv2d_nhwc_16(x, weight, bias=None, act=None, residual=None, stride=1, pad=None, x_lo=None, weight_lo=None, residual_lo=None,
                   out_f32=False, residual_after_act=False, out=None, in_scale=None, res_scale=None, out_state=None):
    """Convolution + bias + residual + activation on the 16-bit MFMA (tlk_conv2d_nhwc_16).  f16 mode: x / weight / residual float16
    channels_last, returns float16.  Split mode (x_lo given): every tensor a (hi, lo) pair of float16 planes, returns (hi, lo).
    out_f32: returns one float32 tensor instead.  bias float32.  out: a float16 channels_last tensor, or a channel slice of one (e.g. this
    layer's part of a concatenation), to write into -- in split mode a (hi, lo) pair of such tensors with equal strides.
    r06, split mode, SCALED planes (tlk_conv2d_nhwc_16s): in_scale / res_scale = the 1-element float32 scale of the input / residual planes (value =
    scale * (hi + lo * 2**-11)); out_state = this layer's 2-element float32 state {scale of the planes it writes, largest |output| recorded}."""
    import torch
    L = lib()
    _bind_conv16(L)
    N, Cin, H, W = x.shape
    Cout, Cw, KH, KW = weight.shape
    if pad is None:
        pad = (KH - 1) // 2
    assert x.dtype == torch.float16 and weight.dtype == torch.float16 and Cw == Cin
    assert bias is None or bias.dtype == torch.float32
    Ho, Wo = (H + 2 * pad - KH) // stride + 1, (W + 2 * pad - KW) // stride + 1
    split = x_lo is not None

    def cl(wt):
        return wt if wt.is_contiguous(memory_format=torch.channels_last) or (KH == 1 and KW == 1 and wt.is_contiguous()) \
            else wt.contiguous(memory_format=torch.channels_last)
    wk, wkl = cl(weight), (cl(weight_lo) if split else None)
    mk = lambda dt: torch.empty((N, Cout, Ho, Wo), dtype=dt, device=x.device, memory_format=torch.channels_last)      # noqa: E731
    y32 = mk(torch.float32) if out_f32 else None
    out_lo = None
    if out is not None:
        assert not out_f32
        if split:
            out, out_lo = out
            assert out_lo.dtype == torch.float16 and out_lo.shape == (N, Cout, Ho, Wo) and _pix16(out_lo, Cout, Ho, Wo) == _pix16(out, Cout, Ho, Wo)
        assert out.dtype == torch.float16 and out.shape == (N, Cout, Ho, Wo)
    yh = None if out_f32 else (out if out is not None else mk(torch.float16))
    yl = (out_lo if out_lo is not None else mk(torch.float16)) if (split and not out_f32) else None
    ptr = lambda t: t.data_ptr() if t is not None else None      # noqa: E731
    scaled = in_scale is not None or res_scale is not None or out_state is not None
    common = (x.data_ptr(), ptr(x_lo), wk.data_ptr(), ptr(wkl), ptr(bias), ptr(residual), ptr(residual_lo),
              ptr(yh), ptr(yl), ptr(y32), N, H, W, Cin, Cout, KH, KW, stride, pad, ACT[act] | (ACT_RES_AFTER if residual_after_act else 0),
              _pix16(x, Cin, H, W), _pix16(out, Cout, Ho, Wo) if out is not None else Cout,
              _pix16(residual, Cout, Ho, Wo) if residual is not None else 0)
    if scaled:
        assert split and all(t is None or (t.dtype == torch.float32 and t.is_cuda) for t in (in_scale, res_scale, out_state))
        assert out_state is None or (out_state.numel() == 2 and out_state.is_contiguous())
        check(L.tlk_conv2d_nhwc_16s(*common, ptr(in_scale), ptr(res_scale), ptr(out_state), current_stream_ptr()))
    else:
        check(L.tlk_conv2d_nhwc_16(*common, current_stream_ptr()))
    if out_f32:
        return y32
    return (yh, yl) if split else yh


def split_planes(x, c_out=None, state=None, dynamic_batch=False):
    """float32 (N, C, H, W) channels_last (or a channel slice of one) -> (hi, lo) float16 planes (N, c_out, H, W) channels_last, zero-padded
    channels: value = hi + lo * 2**-11.  state (r06): 2-element float32 {scale, largest |x| recorded}: the planes hold x / scale.
    dynamic_batch: only the live images of tlk_conv_set_dynamic_batch are converted (and recorded)."""
    import torch
    L = lib()
    _bind_conv16(L)
    N, Cc, H, W = x.shape
    c_out = c_out or Cc
    assert x.dtype == torch.float32
    hi = torch.empty((N, c_out, H, W), dtype=torch.float16, device=x.device, memory_format=torch.channels_last)
    lo = torch.empty_like(hi)
    if state is None and not dynamic_batch:
        check(L.tlk_split_f32_planes(x.data_ptr(), N * H * W, Cc, _pix16(x, Cc, H, W), c_out, hi.data_ptr(), lo.data_ptr(), current_stream_ptr()))
    else:
        assert state is None or (state.dtype == torch.float32 and state.numel() == 2 and state.is_contiguous())
        check(L.tlk_split_f32_planes_s(x.data_ptr(), N * H * W, Cc, _pix16(x, Cc, H, W), c_out, hi.data_ptr(), lo.data_ptr(),
                                       state.data_ptr() if state is not None else None, H * W if dynamic_batch else 0, current_stream_ptr()))
    return hi, lo


def merge_planes(hi, lo, scale=None):
    """hi + lo * 2**-11 (times the planes' 1-element float32 scale, r06) as a float32 channels_last tensor"""
    import torch
    L = lib()
    _bind_conv16(L)
    assert hi.dtype == torch.float16 and lo.dtype == torch.float16 and hi.shape == lo.shape
    assert hi.is_contiguous(memory_format=torch.channels_last) and lo.is_contiguous(memory_format=torch.channels_last)
    y = torch.empty(hi.shape, dtype=torch.float32, device=hi.device, memory_format=torch.channels_last)
    if scale is None:
        check(L.tlk_merge_planes_f32(hi.data_ptr(), lo.data_ptr(), hi.numel(), y.data_ptr(), current_stream_ptr()))
    else:
        assert scale.dtype == torch.float32 and scale.is_cuda
        check(L.tlk_merge_planes_f32_s(hi.data_ptr(), lo.data_ptr(), hi.numel(), scale.data_ptr(), y.data_ptr(), current_stream_ptr()))
    return y


def split_fuse_sum(terms, relu=False, out=None, out_state=None, dynamic_batch=False):
    """``tlk_split_fuse_sum`` (r06): [relu]( sum of terms ) as scaled (hi, lo) planes in one pass.  terms: 1..4 of (hi, lo, scale-or-None) plane
    triples or float32 tensors, channels_last (N, C, H >> s, W >> s) with s >= 0 relative to the output's resolution (H, W) = that of `out`, else the
    largest among the terms; the others are up-sampled by 2**s (nearest).  out: a (hi, lo) pair of float16 channels_last tensors or channel
    slices to write into (default: new tensors); out_state: {scale, recorded maximum} of the planes written.  Returns (hi, lo)."""
    import torch
    L = lib()
    _bind_conv16(L)
    assert 1 <= len(terms) <= 4
    shapes = [(t[0] if isinstance(t, tuple) else t).shape for t in terms]
    N, Cc = shapes[0][0], shapes[0][1]
    H, W = (out[0].shape[2], out[0].shape[3]) if out is not None else (max(s[2] for s in shapes), max(s[3] for s in shapes))
    n = len(terms)
    P, I = C.c_void_p * n, C.c_int * n
    hi_t, lo_t, f_t, sc_t, sh_t, px_t = P(), P(), P(), P(), I(), I()
    keep = []
    for i, t in enumerate(terms):
        shp = shapes[i]
        assert shp[0] == N and shp[1] == Cc, "terms must agree in batch and channels"
        s = (H // shp[2]).bit_length() - 1
        assert shp[2] << s == H and shp[3] << s == W, "a term's resolution must divide the output's by a power of two"
        sh_t[i] = s
        if isinstance(t, tuple):
            h_, l_, sc = t
            assert h_.dtype == torch.float16 and l_.dtype == torch.float16 and h_.shape == l_.shape
            px_t[i] = _pix16(h_, Cc, shp[2], shp[3])
            assert _pix16(l_, Cc, shp[2], shp[3]) == px_t[i]
            hi_t[i], lo_t[i], f_t[i] = h_.data_ptr(), l_.data_ptr(), None
            sc_t[i] = sc.data_ptr() if sc is not None else None
        else:
            assert t.dtype == torch.float32
            px_t[i] = _pix16(t, Cc, shp[2], shp[3])
            hi_t[i], lo_t[i], f_t[i], sc_t[i] = None, None, t.data_ptr(), None
        keep.append(t)
    if out is None:
        yh = torch.empty((N, Cc, H, W), dtype=torch.float16, device=keep[0][0].device if isinstance(keep[0], tuple) else keep[0].device,
                         memory_format=torch.channels_last)
        yl = torch.empty_like(yh)
    else:
        yh, yl = out
        assert yh.dtype == torch.float16 and yl.dtype == torch.float16 and tuple(yh.shape) == (N, Cc, H, W) and yl.shape == yh.shape
        assert _pix16(yh, Cc, H, W) == _pix16(yl, Cc, H, W)
    assert out_state is None or (out_state.dtype == torch.float32 and out_state.numel() == 2 and out_state.is_contiguous())
    check(L.tlk_split_fuse_sum(n, hi_t, lo_t, f_t, sc_t, sh_t, px_t, N, H, W, Cc, 1 if relu else 0, yh.data_ptr(), yl.data_ptr(), _pix16(yh, Cc, H, W),
                               out_state.data_ptr() if out_state is not None else None, 1 if dynamic_batch else 0, current_stream_ptr()))
    return yh, yl


def fuse_sum_f32(terms, relu=False, out=None, dynamic_batch=False):
    """``tlk_fuse_sum_f32`` (r06): [relu](((t0 + t1) + t2) + t3) over float32 channels_last tensors at mixed resolutions (nearest up-sampling by
    2**s onto the output's grid = that of `out`, else the largest term's), bit-identical to torch's interpolate / add / relu composition, one pass.
    out: a float32 channels_last tensor or channel slice to write into."""
    import torch
    return _fuse_sum_plain(terms, relu, out, dynamic_batch, torch.float32)


def fuse_sum_f16(terms, relu=False, out=None, dynamic_batch=False):
    """``tlk_fuse_sum_f16``: the same over float16 tensors, every partial sum rounded to float16 (torch's half-precision `y = y + t` chain, bit for bit)"""
    import torch
    return _fuse_sum_plain(terms, relu, out, dynamic_batch, torch.float16)


def _fuse_sum_plain(terms, relu, out, dynamic_batch, dtype):
    import torch
    L = lib()
    _bind_conv16(L)
    assert 1 <= len(terms) <= 4
    N, Cc = terms[0].shape[0], terms[0].shape[1]
    H, W = (out.shape[2], out.shape[3]) if out is not None else (max(t.shape[2] for t in terms), max(t.shape[3] for t in terms))
    n = len(terms)
    P, I = C.c_void_p * n, C.c_int * n
    x_t, sh_t, px_t = P(), I(), I()
    for i, t in enumerate(terms):
        assert t.dtype == dtype and t.shape[0] == N and t.shape[1] == Cc, "terms must share the dtype and agree in batch and channels"
        s = (H // t.shape[2]).bit_length() - 1
        assert t.shape[2] << s == H and t.shape[3] << s == W, "a term's resolution must divide the output's by a power of two"
        x_t[i], sh_t[i], px_t[i] = t.data_ptr(), s, _pix16(t, Cc, t.shape[2], t.shape[3])
    if out is None:
        out = torch.empty((N, Cc, H, W), dtype=dtype, device=terms[0].device, memory_format=torch.channels_last)
    assert out.dtype == dtype and tuple(out.shape) == (N, Cc, H, W)
    fn = L.tlk_fuse_sum_f32 if dtype == torch.float32 else L.tlk_fuse_sum_f16
    check(fn(n, x_t, sh_t, px_t, N, H, W, Cc, 1 if relu else 0, out.data_ptr(), _pix16(out, Cc, H, W), 1 if dynamic_batch else 0, current_stream_ptr()))
    return out


def split_scale_update(states, changed=None):
    """``tlk_split_scale_update``: states (n, 2) float32 {scale, recorded maximum} -> the scales of the NEXT forward (powers of two >= 1; growth at
    once, shrinking with hysteresis), maxima cleared; changed (1,) int32 is incremented per state whose scale grew or whose maximum was not finite."""
    import torch
    L = lib()
    _bind_conv16(L)
    assert states.dtype == torch.float32 and states.is_contiguous() and states.shape[-1] == 2
    assert changed is None or (changed.dtype == torch.int32 and changed.numel() == 1)
    check(L.tlk_split_scale_update(states.data_ptr(), states.numel() // 2, changed.data_ptr() if changed is not None else None, current_stream_ptr()))


def cosine_gallery_min(gallery, offsets, dets):
    """gallery (G, D) f32, offsets (T+1,) int32 (CSR per track), dets (N, D) f32 cuda tensors -> (T, N) f64."""
    import torch
    L = lib()
    if not getattr(L, "_cos_bound", False):
        L.tlk_cosine_gallery_min_f32.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int,
                                                 C.c_void_p, C.c_void_p]
        L._cos_bound = True
    T, (N, D) = offsets.numel() - 1, dets.shape
    assert gallery.dtype == torch.float32 and dets.dtype == torch.float32 and offsets.dtype == torch.int32
    assert gallery.is_contiguous() and dets.is_contiguous()
    out = torch.empty((T, N), dtype=torch.float64, device=dets.device)
    check(L.tlk_cosine_gallery_min_f32(gallery.data_ptr(), offsets.data_ptr(), T, gallery.shape[0], dets.data_ptr(), N, D,
                                       out.data_ptr(), current_stream_ptr()))
    return out


def pyset_difference_order(a, b, force_table=False):
    """`list(set(a) - set(b))` in CPython 3.10's set-iteration order, computed on the device by the code the StrongSORT-family
    association kernels run (tlk_pyset.hpp; sort/linear_assignment.py:126-128). a ascending distinct non-negative ints, b out of a."""
    L = lib()
    ip = C.POINTER(C.c_int32)
    L.tlk_pyset_difference_order.argtypes = [ip, C.c_int, ip, C.c_int, ip, C.POINTER(C.c_int32), C.c_int]
    a = np.ascontiguousarray(a, dtype=np.int32)
    b = np.ascontiguousarray(b, dtype=np.int32)
    out = np.zeros(max(len(a), 1), dtype=np.int32)
    n = C.c_int32(0)
    check(L.tlk_pyset_difference_order(a.ctypes.data_as(ip), len(a), b.ctypes.data_as(ip), len(b), out.ctypes.data_as(ip), C.byref(n),
                                       int(bool(force_table))))
    return out[:n.value].copy()


# ------------------------------------------------------------------------------------------------
# camera-motion estimation on the device (tlk_cmc_*): GMC.applySparseOptFlow of BoT-SORT (gmc.py:239-303)
# ------------------------------------------------------------------------------------------------
def deepsort_nms(boxes, max_bbox_overlap, scores=None):
    """sort/preprocessing.py:6-73 non_max_suppression on the device (tlk_deepsort_nms_f64): cuda tensors (n, 4) float64 xywh [+ (n,) float64
    scores] -> int32 cuda tensor of the kept indices, in pick order."""
    import torch
    assert boxes.is_cuda and boxes.dtype == torch.float64 and boxes.is_contiguous() and boxes.dim() == 2 and boxes.shape[1] == 4
    n = boxes.shape[0]
    pick = torch.zeros(max(n, 1), dtype=torch.int32, device=boxes.device)
    cnt = torch.zeros(1, dtype=torch.int32, device=boxes.device)
    sp = None
    if scores is not None:
        assert scores.is_cuda and scores.dtype == torch.float64 and scores.is_contiguous() and scores.numel() == n
        sp = scores.data_ptr()
    L = lib()
    L.tlk_deepsort_nms_f64.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_double, C.c_void_p, C.c_void_p, C.c_void_p]
    check(L.tlk_deepsort_nms_f64(boxes.data_ptr() if n else None, sp, n, float(max_bbox_overlap), pick.data_ptr(), cnt.data_ptr(), current_stream_ptr()))
    return pick[:int(cnt.item())]


class EccEstimator:
    """One video stream's StrongSORT camera-motion estimator (Track.ECC, strong_sort/sort/track.py:129-211): ``apply(frame)`` -> the
    (2, 3) float32 warp for ``SsortBank.camera_update``, or None where the reference skips the update (first frame; cv2.error).
    ``apply_dev(frame_cuda_tensor)`` leaves warp (6 doubles) and status (int32) in device memory. OpenCV's findTransformECC restated:
    PARITY UNPINNED (tlk_ecc.hip, oracle/src/ecc.c)."""

    def __init__(self, height, width, device=0):
        L = lib()
        vp, ci = C.c_void_p, C.c_int
        L.tlk_ecc_create.argtypes = [ci, ci, ci, C.POINTER(vp)]
        L.tlk_ecc_destroy.argtypes = [vp]
        L.tlk_ecc_reset.argtypes = [vp]
        L.tlk_ecc_apply_dev.argtypes = [vp, vp, vp, vp, vp]
        L.tlk_ecc_apply.argtypes = [vp, vp, vp, C.POINTER(ci), C.POINTER(C.c_double)]
        self.h, self.w = int(height), int(width)
        h = vp()
        check(L.tlk_ecc_create(self.h, self.w, device, C.byref(h)))
        self._h = h
        self.iterations, self.rho = 0, 0.0
        self.warp_dev = self.status_dev = None

    def close(self):
        if getattr(self, "_h", None):
            lib().tlk_ecc_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def reset(self):
        check(lib().tlk_ecc_reset(self._h))

    def apply(self, frame):
        frame = np.ascontiguousarray(frame, dtype=np.uint8)
        assert frame.shape == (self.h, self.w, 3)
        H = np.zeros(6)
        n, rho = C.c_int(0), C.c_double(0.0)
        check(lib().tlk_ecc_apply(self._h, frame.ctypes.data, H.ctypes.data, C.byref(n), C.byref(rho)))
        self.iterations, self.rho = n.value, rho.value
        return H.reshape(2, 3).astype(np.float32) if n.value >= 1 else None

    def apply_dev(self, frame, stream_ptr=None):
        import torch
        assert frame.is_cuda and frame.dtype == torch.uint8 and frame.is_contiguous() and tuple(frame.shape) == (self.h, self.w, 3)
        if self.warp_dev is None:
            self.warp_dev = torch.zeros(6, dtype=torch.float64, device=frame.device)
            self.status_dev = torch.zeros(1, dtype=torch.int32, device=frame.device)
        check(lib().tlk_ecc_apply_dev(self._h, frame.data_ptr(), self.warp_dev.data_ptr(), self.status_dev.data_ptr(),
                                      stream_ptr if stream_ptr is not None else current_stream_ptr()))
        return self.warp_dev, self.status_dev


def ecc_find_transform(templ, image, max_iter=100, eps=1e-5, device=0):
    """cv2.findTransformECC (Euclidean, from the identity) alone on two (h, w) uint8 images -> (warp (2,3) float32, iterations | -1, rho)."""
    templ = np.ascontiguousarray(templ, dtype=np.uint8); image = np.ascontiguousarray(image, dtype=np.uint8)
    assert templ.shape == image.shape and templ.ndim == 2
    L = lib()
    L.tlk_ecc_find_transform.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_double, C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_double), C.c_int]
    H = np.zeros(6)
    n, rho = C.c_int(0), C.c_double(0.0)
    check(L.tlk_ecc_find_transform(templ.ctypes.data, image.ctypes.data, templ.shape[0], templ.shape[1], int(max_iter), float(eps), H.ctypes.data, C.byref(n),
                                   C.byref(rho), device))
    return H.reshape(2, 3).astype(np.float32), n.value, rho.value


class CmcEstimator:
    """One video stream's camera-motion estimator: ``apply(frame)`` -> (2, 3) float64 warp for ``BoTSORTBank.update(..., warp=)``;
    ``apply_dev(frame_cuda_tensor)`` leaves the warp in device memory (``.warp_dev``: 6 doubles) for ``update_dev(..., warps=)`` with
    no host synchronisation. OpenCV's arithmetic restated: PARITY UNPINNED (tlk_cmc.hip, oracle/src/cmc.c)."""

    def __init__(self, height, width, downscale=2, max_corners=1000, device=0):
        L = lib()
        vp, ci = C.c_void_p, C.c_int
        L.tlk_cmc_create.argtypes = [ci, ci, ci, ci, ci, C.POINTER(vp)]
        L.tlk_cmc_destroy.argtypes = [vp]
        L.tlk_cmc_reset.argtypes = [vp]
        L.tlk_cmc_apply_dev.argtypes = [vp, vp, vp, vp]
        L.tlk_cmc_apply_dev_gated.argtypes = [vp, vp, vp, vp, vp]
        L.tlk_cmc_apply.argtypes = [vp, vp, vp, C.POINTER(ci)]
        L.tlk_cmc_debug_get.argtypes = [vp, ci, vp, C.c_size_t, C.POINTER(ci)]
        self.h, self.w, self.downscale = int(height), int(width), max(1, int(downscale))
        self.dh, self.dw = (self.h // self.downscale, self.w // self.downscale) if self.downscale > 1 else (self.h, self.w)
        h = vp()
        check(L.tlk_cmc_create(self.h, self.w, self.downscale, int(max_corners), device, C.byref(h)))
        self._h = h
        self.inliers = 0
        self.warp_dev = None

    def close(self):
        if getattr(self, "_h", None):
            lib().tlk_cmc_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def reset(self):
        check(lib().tlk_cmc_reset(self._h))

    def apply(self, frame):
        frame = np.ascontiguousarray(frame, dtype=np.uint8)
        assert frame.shape == (self.h, self.w, 3)
        H = np.zeros(6)
        n = C.c_int(0)
        check(lib().tlk_cmc_apply(self._h, frame.ctypes.data, H.ctypes.data, C.byref(n)))
        self.inliers = n.value
        return H.reshape(2, 3)

    def apply_dev(self, frame, stream_ptr=None, out=None, count=None):
        """out: optional (6,) float64 cuda tensor (e.g. a row of the (S, F, 6) warps block of ``BoTSORTBank.update_dev``) instead of ``.warp_dev``.
        count: optional 1-element int32 cuda tensor = the frame's detection count; when it is 0 the estimator's state is rolled back on the device
        (the reference does not call GMC.apply on a frame without detections, bot_sort_api.py:59-60): tlk_cmc_apply_dev_gated."""
        import torch
        assert frame.is_cuda and frame.dtype == torch.uint8 and frame.is_contiguous() and tuple(frame.shape) == (self.h, self.w, 3)
        if out is None:
            if self.warp_dev is None:
                self.warp_dev = torch.zeros(6, dtype=torch.float64, device=frame.device)
            out = self.warp_dev
        assert out.is_cuda and out.dtype == torch.float64 and out.numel() == 6 and out.is_contiguous()
        sp = stream_ptr if stream_ptr is not None else current_stream_ptr()
        if count is None:
            check(lib().tlk_cmc_apply_dev(self._h, frame.data_ptr(), out.data_ptr(), sp))
        else:
            assert count.is_cuda and count.dtype == torch.int32 and count.numel() == 1
            check(lib().tlk_cmc_apply_dev_gated(self._h, frame.data_ptr(), out.data_ptr(), count.data_ptr(), sp))
        return out

    def debug(self, what):
        """Stage outputs of the last apply (tlk.h tlk_cmc_debug_get)."""
        n = C.c_int(0)
        if what in (0,) or 10 <= what < 20:
            buf = np.zeros(self.dh * self.dw, np.uint8)
        elif what == 1:
            buf = np.zeros(self.dh * self.dw, np.float32)
        elif what in (2, 3):
            buf = np.zeros((1024, 2), np.float32)
        elif what == 4:
            buf = np.zeros(1024, np.uint8)
        else:
            buf = np.zeros(self.dh * self.dw * 2, np.int16)
        check(lib().tlk_cmc_debug_get(self._h, what, buf.ctypes.data, buf.nbytes, C.byref(n)))
        if what == 0:
            return buf.reshape(self.dh, self.dw)
        if what == 1:
            return buf.reshape(self.dh, self.dw)
        if what in (2, 3):
            return buf[:n.value].copy()
        if what == 4:
            return buf[:n.value].astype(bool)
        w = n.value
        if 10 <= what < 20:
            h = -(-self.dh // (1 << (what - 10)))
            return buf[:h * w].reshape(h, w) if what == 10 else buf[:(buf.size // w) * w].reshape(-1, w)
        return buf[:(buf.size // (2 * w)) * 2 * w].reshape(-1, w, 2)
