"""GPU-resident per-frame loops: detector -> (ReID ->) association with no host round trip.

This is the data plane the north star describes: frames sit in HBM, every stage runs on ONE HIP
stream (torch's current stream), libtlk kernels do everything except the backbone forwards, and the
only device->host traffic is the small per-step result block (async, pinned).

Channel contract of ``step(frames)``: frames are RGB, as TrackLab's ``cv2_load_image`` hands them to every module. The reference's
detector and pose estimator re-read the file with ``cv2.imread`` (BGR; wrappers/bbox_detector/rtmlib_api.py:30,
wrappers/pose_estimator/rtmlib_api.py:30) while the ReID crops are cut from the RGB image and normalised with RGB ImageNet statistics
(wrappers/reid/kpreid_api.py:115-144; strong_sort/reid_multibackend.py:184-195). So the letterbox and the pose warp read the frame with
``TLK_SWAP_RB`` and the ReID crop kernels read it as is -- one copy of the frame in HBM, no flip pass.

``DetTrackPipeline``  = BASELINE.json configs[1]: YOLOX -> OC-SORT (no ReID).
A *step* processes ``frames_per_step`` consecutive frames of each of ``n_streams`` streams:
  letterbox (1 launch) -> YOLOX forward (torch/MIOpen) -> decode+NMS (1 launch, emits tracker rows)
  -> OC-SORT (1 launch: one workgroup per stream walks its frames in order).
"""
from __future__ import annotations

import ctypes as C

import numpy as np
import torch

from . import _lib
from .backbones.yolox import yolox


def streams_run_concurrently(sa, sb, cycles: int = 40_000, chain: int = 16) -> bool:
    """Do two streams execute side by side?  torch hands out its side streams from a pool (32 per priority, round robin) and HIP deals each of
    them -- at its creation -- to one of a few hardware queues (GPU_MAX_HW_QUEUES, 4 by default); two streams on ONE queue run strictly one after
    the other whatever the events between them say.  Which pool stream a `torch.cuda.Stream()` call returns depends on how many were drawn
    before it in the process, so a pipeline's association / copy / stage stream lands on the compute stream's queue about one time in four:
    measured r06 as a config-2 step of 17.2 instead of 12.3 ms, an f16 step of 54.7 instead of 50.3 ms, an H2D-inclusive leg 3 % behind the
    resident one, the overlapped online pipeline at 0.5 x the serial one (profiles/r06_overlap_autotune.md).  Measured here with the launch
    pattern of real work: a CHAIN of dependent spin kernels on each stream, submitted interleaved, timestamps on one clock."""
    dev = sa.device
    if sa is sb or sa.cuda_stream == sb.cuda_stream:
        return False

    def run(streams, n):
        """`n` spin kernels per stream, submitted interleaved; wall time from the first start to the last end (events on one clock)"""
        e0, e1 = torch.cuda.Event(enable_timing=True), [torch.cuda.Event(enable_timing=True) for _ in streams]
        torch.cuda.synchronize(dev)
        cur = torch.cuda.current_stream(dev)
        e0.record(cur)
        for s_ in streams:
            s_.wait_event(e0)
        for k in range(n):
            for i, s_ in enumerate(streams):
                with torch.cuda.stream(s_):
                    torch.cuda._sleep(cycles)
                    if k == n - 1:
                        e1[i].record(s_)
        for s_ in streams:
            s_.synchronize()
        return max(e0.elapsed_time(e) for e in e1)
    run([sa, sb], 2)                          # warm (module load of the spin kernel)
    t_one = min(run([sa], chain), run([sb], chain))
    t_both = run([sa, sb], chain)
    # side by side: both chains take about as long as one; one queue (or time-sliced queues): about twice as long
    return t_both < 1.45 * t_one


_PICK_LOG = []


def pick_stream(dev, others, priority: int = 0, tries: int = 6):
    """A side stream that was MEASURED to run concurrently with every stream in `others` (see streams_run_concurrently): draws from torch's pool
    until one does (each draw is the next pool stream, i.e. usually the next hardware queue); after `tries` failures the first draw is returned
    and the caller's work simply serialises.  TLK_PICK_STREAMS=0: the first draw, unmeasured (r01-r05)."""
    first = None
    measured = __import__("os").environ.get("TLK_PICK_STREAMS", "1") != "0"
    for k in range(tries):
        s_ = torch.cuda.Stream(device=dev, priority=priority)
        first = first or s_
        # (pairwise on purpose: a set of three or four streams never passed the all-at-once form of the measurement on this chip, and with pairwise
        #  picks the forced stage overlap ran at 1.40 x the serial pipeline in 9 of 9 process states against 2 of 9 with first draws)
        if not measured or all(streams_run_concurrently(s_, o) for o in others if o is not None):
            _PICK_LOG.append((k, True))
            return s_
    _PICK_LOG.append((tries, False))
    return first


def _record_null_pair(self):
    """Two timing events with nothing between them on the current stream: the cost of an event pair itself (a few us on this stack -- not negligible
    beside a 30 us kernel). bench.py subtracts its mean from the mean of the kernel's own event pairs and prints both."""
    n0, n1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n0.record(); n1.record()
    self.null_events.append((n0, n1))


class DetTrackPipeline:
    _record_null_pair = _record_null_pair
    def __init__(self, detector: str = "s", n_streams: int = 1, frames_per_step: int = 16, max_dets: int = 128,
                 height: int = 1080, width: int = 1920, size: int = 640, dtype=torch.float16,
                 layout: str = "focus_nhwc", device: int = 0, tracker_cfg: dict | None = None,
                 num_classes: int = 1, nms_thr: float = 0.45, score_thr: float = 0.7, max_tracks: int = 256,
                 use_graph: bool = True, tracker: str = "oc_sort"):
        """tracker: "oc_sort" (configs[1]) or "byte_track" (same detector, ByteTrack association)."""
        self.tracker = tracker
        dtype = getattr(torch, dtype) if isinstance(dtype, str) else dtype        # yaml: "float16" / "float32"
        self.S, self.F, self.maxd = n_streams, frames_per_step, max_dets
        self.H, self.W, self.size, self.dtype, self.layout = height, width, size, dtype, layout
        self.nms_thr, self.score_thr = nms_thr, score_thr
        self.dev = torch.device("cuda", device)
        torch.cuda.set_device(self.dev)
        cfg = tracker_cfg or dict(
            # tracklab/configs/modules/track/oc_sort.yaml:3-14
            min_confidence=0.4, hyper=dict(asso_func="giou", delta_t=1, det_thresh=0, inertia=0.3941737016672115,
                                           iou_threshold=0.22136877277096445, max_age=50, min_hits=1, use_byte=False))
        if tracker == "byte_track" and tracker_cfg is None:      # tracklab/configs/modules/track/byte_track.yaml
            cfg = dict(min_confidence=0.4, hyper=dict(track_thresh=0.6, track_buffer=30, match_thresh=0.8, frame_rate=30))
        self.tracker_cfg = cfg
        self.model = yolox(detector, num_classes, device=self.dev, dtype=dtype, channels_last=(layout != "nchw"))
        if tracker == "byte_track":
            self.bank = _lib.ByteTrackBank(**cfg["hyper"], min_confidence=cfg["min_confidence"], wrapper_mode=True,
                                           n_streams=n_streams, device=device, max_tracks=max_tracks, max_dets=max_dets)
            self.row_dtype = _lib.BYTETRACK_ROW
        else:
            self.bank = _lib.OCSortBank(**cfg["hyper"], min_confidence=cfg["min_confidence"], wrapper_mode=True,
                                        n_streams=n_streams, device=device, max_tracks=max_tracks, max_dets=max_dets)
            self.row_dtype = None
        B = n_streams * frames_per_step
        self.B = B
        A = sum((size // s) ** 2 for s in (8, 16, 32))
        self.A = A
        dev = self.dev
        if layout == "nchw":
            self.lb = torch.empty((B, 3, size, size), dtype=dtype, device=dev)
        elif layout == "nhwc":
            self.lb = torch.empty((B, size, size, 3), dtype=dtype, device=dev)
        else:
            self.lb = torch.empty((B, size // 2, size // 2, 12), dtype=dtype, device=dev)
        self.out_cap = max_dets
        self.nbuf = 2       # double-buffered hand-off between the detector stream and the tracker stream
        self.bufs = []
        for _ in range(self.nbuf):
            self.bufs.append({
                "det": {"ltwh": torch.zeros((B, max_dets, 4), dtype=torch.float32, device=dev),
                        "xyxy": torch.zeros((B, max_dets, 4), dtype=torch.float32, device=dev),
                        "scores": torch.zeros((B, max_dets), dtype=torch.float32, device=dev),
                        "cls": torch.zeros((B, max_dets), dtype=torch.int32, device=dev),
                        "counts": torch.zeros((B,), dtype=torch.int32, device=dev)},
                "trk_in": torch.zeros((n_streams, frames_per_step, max_dets, 7), dtype=torch.float64, device=dev),
                "trk_out": torch.zeros((n_streams, frames_per_step, self.out_cap, 8), dtype=torch.float64, device=dev),
                "trk_cnt": torch.zeros((n_streams, frames_per_step), dtype=torch.int32, device=dev),
                "h_out": torch.zeros((n_streams, frames_per_step, self.out_cap, 8), dtype=torch.float64).pin_memory(),
                "h_cnt": torch.zeros((n_streams, frames_per_step), dtype=torch.int32).pin_memory(),
                "h_ltwh": torch.zeros((B, max_dets, 4), dtype=torch.float32).pin_memory(),
                "h_dcnt": torch.zeros((B,), dtype=torch.int32).pin_memory(),
                "det_ready": torch.cuda.Event(), "trk_done": torch.cuda.Event()})
        # association overlaps the next step (high priority measured slower: 227 vs 262 frames/s); r06: a stream MEASURED to run beside the compute stream
        self.trk_stream = pick_stream(dev, [torch.cuda.current_stream(dev)], priority=int(__import__("os").environ.get("TLK_TRK_PRIO", "0")))
        self.use_graph = use_graph
        self.graphs = {}        # frames.data_ptr() -> (hipGraph of letterbox + forward, static head output)
        self.step_idx = 0
        self.ratio = min(size / height, size / width)
        self.frames_done = 0
        self.kernel_events = []         # (start, end) torch events around the letterbox launch
        self.null_events = []           # (start, end) with nothing between them, recorded just before: what an event pair itself costs on this stream
        self.record_kernel_events = False

    def reset(self, keep_ids: bool = False):
        """keep_ids: ByteTrack / BoT-SORT keep their id counter (the reference's class-level BaseTrack._count); the other trackers' counters are
        re-created with the tracker in the reference, so they restart at 1 either way."""
        torch.cuda.synchronize(self.dev)
        if keep_ids and self.tracker in ("byte_track", "bot_sort"):
            self.bank.reset(-1, keep_ids=True)
        else:
            self.bank.reset(-1)
        for est in (getattr(self, "cmc", None) or []):
            est.reset()
        self.frames_done = 0

    def synchronize(self):
        self.trk_stream.synchronize()
        torch.cuda.current_stream(self.dev).synchronize()

    @torch.no_grad()
    def step(self, frames: torch.Tensor, synth_head: torch.Tensor | None = None, fetch: bool = True, sink=None):
        """frames: (S*F, H, W, 3) uint8 RGB on device, ordered stream-major (s*F + f).
        sink: optional callable(result_tensors) run on the association stream right after the tracker launch (device-side
        consumers such as the engine's HBM-resident detection table: device-to-device copies, no host sync).
        After the call ``self.frames_free`` is an event on the main stream behind the last kernel that reads ``frames``.
        synth_head: optional (S*F, A, 5+C) float32 replacing the (random-init) detector's head activations
        while keeping the full forward in the dependency chain. Returns (rows, counts) pinned host tensors
        (valid after ``synchronize()``) or device tensors if fetch=False.
        Detector stages run on torch's current stream; the association kernel + result copies run on a side
        stream, so the tracker of step k overlaps the detector forward of step k+1."""
        S, F = self.S, self.F
        buf = self.bufs[self.step_idx % self.nbuf]
        self.step_idx += 1
        main = torch.cuda.current_stream(self.dev)
        main.wait_event(buf["trk_done"])      # buffer reuse: tracker of step k-nbuf has consumed it
        if self.record_kernel_events:
            self._record_null_pair()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
        ratio = self.ratio
        if self.use_graph and not self.record_kernel_events:
            pred = self._forward_graphed(frames)
        else:
            x, ratio = _lib.letterbox(frames, self.size, self.layout, self.dtype, out=self.lb, swap_rb=True)
            if self.record_kernel_events:
                e1.record()
                self.kernel_events.append((e0, e1))
            pred = self.model(x, focused=(self.layout == "focus_nhwc"))
        self.frames_free = torch.cuda.Event()
        self.frames_free.record(main)
        if synth_head is not None:
            pred = torch.add(synth_head, torch.nan_to_num(pred), alpha=0.0)
        _lib.yolox_decode_nms(pred, self.size, float(np.float32(ratio)), self.W, self.H, self.maxd, self.nms_thr,
                              self.score_thr, out=buf["det"], trk_in=buf["trk_in"],
                              det_id_base=self.frames_done * self.maxd, category_id=1.0)
        buf["det_ready"].record(main)
        self.frames_done += S * F
        with torch.cuda.stream(self.trk_stream):
            self.trk_stream.wait_event(buf["det_ready"])
            self.bank.update_dev(buf["trk_in"].data_ptr(), buf["det"]["counts"].data_ptr(), F, buf["trk_out"].data_ptr(),
                                 self.out_cap, buf["trk_cnt"].data_ptr(), C.c_void_p(self.trk_stream.cuda_stream))
            if sink is not None:
                sink(self.result_tensors(buf))
            if fetch:       # every host-visible result of the step is in pinned memory before `trk_done`
                buf["h_out"].copy_(buf["trk_out"], non_blocking=True)
                buf["h_cnt"].copy_(buf["trk_cnt"], non_blocking=True)
                buf["h_ltwh"].copy_(buf["det"]["ltwh"], non_blocking=True)
                buf["h_dcnt"].copy_(buf["det"]["counts"], non_blocking=True)
            buf["trk_done"].record(self.trk_stream)
        self.last = buf
        if not fetch:
            return buf["trk_out"], buf["trk_cnt"]
        return buf["h_out"], buf["h_cnt"]

    def result_tensors(self, buf):
        """Device tensors one step leaves behind, in the shape the engine's detection table stores them (n_streams == 1: B == F)."""
        return {"rows": buf["trk_out"].view(self.B, self.out_cap, 8), "ocnt": buf["trk_cnt"].view(self.B),
                "ltwh": buf["det"]["ltwh"], "dcnt": buf["det"]["counts"]}

    def host_results(self, buf):
        """The pinned host copies of the same four arrays (valid once ``buf["trk_done"]`` has been synchronised)."""
        return {"rows": buf["h_out"].numpy().reshape(self.B, self.out_cap, 8), "ocnt": buf["h_cnt"].numpy().reshape(self.B),
                "ltwh": buf["h_ltwh"].numpy(), "dcnt": buf["h_dcnt"].numpy()}

    done_key = "trk_done"

    def track_columns(self, rows, ocnt):
        """(frames, cap, 8) float64 row blocks + (frames,) counts -> flat (frame index, det_id, track_id, ltwh, conf) of the valid rows."""
        a = self.rows_array_np(rows)
        f, i = np.nonzero(np.arange(a.shape[1])[None, :] < ocnt[:, None])
        r = a[f, i]
        return f, r[:, 7].astype(np.int64), r[:, 4], np.stack([r[:, 0], r[:, 1], r[:, 2] - r[:, 0], r[:, 3] - r[:, 1]], axis=1).reshape(-1, 4), r[:, 6]

    def rows_array_np(self, a):
        if self.row_dtype is None:
            return a
        r = np.ascontiguousarray(a).view(self.row_dtype).reshape(a.shape[:-1])
        out = np.empty(a.shape[:-1] + (8,))
        out[..., :4] = r["ltrb"]; out[..., 4] = r["track_id"]; out[..., 5] = r["cls"]; out[..., 6] = r["score"]; out[..., 7] = r["det_id"]
        return out

    def rows_array(self, h_out):
        """Host result block -> (S, F, cap, 8) float64 rows [x1,y1,x2,y2,track_id,cls,conf,det_id] for either tracker."""
        a = h_out.numpy()
        if self.row_dtype is None:
            return a
        r = a.view(self.row_dtype).reshape(a.shape[:3])          # ByteTrack rows are 8 x 8 bytes as well
        out = np.empty(a.shape[:3] + (8,))
        out[..., :4] = r["ltrb"]; out[..., 4] = r["track_id"]; out[..., 5] = r["cls"]; out[..., 6] = r["score"]; out[..., 7] = r["det_id"]
        return out

    def _forward_graphed(self, frames):
        """letterbox + YOLOX forward replayed from a hipGraph (the forward is ~300 short launches and is
        host-launch-bound in eager mode). One graph per input buffer address."""
        key = frames.data_ptr()
        ent = self.graphs.get(key)
        if ent is None:
            focused = self.layout == "focus_nhwc"
            side = torch.cuda.Stream(device=self.dev)
            side.wait_stream(torch.cuda.current_stream(self.dev))
            with torch.cuda.stream(side):          # eager warm-up on a side stream (MIOpen picks its kernels here)
                for _ in range(2):
                    x, _ = _lib.letterbox(frames, self.size, self.layout, self.dtype, out=self.lb, swap_rb=True)
                    self.model(x, focused=focused)
            torch.cuda.current_stream(self.dev).wait_stream(side)
            torch.cuda.synchronize(self.dev)
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                x, _ = _lib.letterbox(frames, self.size, self.layout, self.dtype, out=self.lb, swap_rb=True)
                out = self.model(x, focused=focused)
            ent = (g, out)
            self.graphs[key] = ent
        ent[0].replay()
        return ent[1]

    def close(self):
        self.graphs.clear()
        self.bank.close()


class DetReidTrackPipeline:
    _record_null_pair = _record_null_pair
    """BASELINE.json configs[2]/[4]: YOLOX -> part-based ReID -> BPBReID-StrongSORT, GPU-resident.

    A step = ``frames_per_step`` consecutive frames of each of ``n_streams`` streams:
      letterbox + YOLOX forward (hipGraph) -> decode+NMS -> ROI crop-resize-normalize of every detection straight
      from the frames in HBM (1 launch) -> ReID forward (hipGraph) -> per frame: part-norms, MFMA part distance,
      association (3 launches) on a side stream that overlaps the next step's detector/ReID forwards.
    """

    def __init__(self, detector: str = "m", n_streams: int = 1, frames_per_step: int = 8, max_dets: int = 104,
                 height: int = 1080, width: int = 1920, size: int = 640, dtype=torch.float16, device: int = 0,
                 parts: int = 6, dim: int = 256, reid_hw=(384, 128), tracker_cfg: dict | None = None,
                 nms_thr: float = 0.45, score_thr: float = 0.7, max_tracks: int | None = None, use_graph: bool = True,
                 pose: str | None = None, tracker: str = "bpbreid", reid_arch: str = "resnet50", camera_motion: bool = False,
                 reid_split_precision: bool = False, overlap_stages: bool | str | None = None, detector_split_precision: bool = False):
        """overlap_stages (r05, default off; env TLK_PIPE_OVERLAP=1): detector stage of step t + 1 and ReID stage of step t on two streams
        (double-buffered crops): what the ONLINE configuration wants, where a one-frame step's kernels have fewer tiles than the chip has CUs.
        camera_motion (tracker "bot_sort"): the reference's default cmc_method sparseOptFlow (configs/modules/track/bot_sort.yaml) -- one
        estimator per stream (tlk_cmc_*) runs over the step's frames on a side stream under the backbone forwards, its (2,3) warps go to the
        tracker's frame kernel in device memory (tlk_botsort_update_dev_gmc).
        reid_arch: "resnet50" (default) or "hrnet32" (the backbone tracklab/configs/modules/reid/bpbreid.yaml:53 names).
        tracker = "strong_sort": plain StrongSORT (strong_sort.StrongSORT: Pillow-semantics 256x128 crops of the int-truncated
        boxes, one global 512-d feature per crop, cosine gallery on MFMA, tlk_ssort bank) instead of BPBReID-StrongSORT.
        pose = "t"/"s"/"m"/"l": BASELINE.json configs[3] -- a top-down RTMPose stage (tlk_pose_crop_warp_norm -> network ->
        tlk_simcc_decode) between detector and ReID; its keypoints drive the tracker's OKS motion cost (motion_criterium "oks")."""
        from .backbones.reid import part_based_reid
        self.tracker = tracker
        dtype = getattr(torch, dtype) if isinstance(dtype, str) else dtype        # yaml: "float16" / "float32"
        # trackers that own a global-feature ReID net on Pillow-semantics 256x128 crops (ReIDDetectMultiBackend): same stages, other bank
        self.global_feat = tracker in ("strong_sort", "bot_sort", "deep_oc_sort")
        if self.global_feat:
            parts, dim, reid_hw = 1, 512, (256, 128)
        self.S, self.F, self.maxd = n_streams, frames_per_step, max_dets
        self.H, self.W, self.size, self.dtype = height, width, size, dtype
        self.K, self.D, self.reid_hw = parts, dim, reid_hw
        self.nms_thr, self.score_thr = nms_thr, score_thr
        self.dev = torch.device("cuda", device)
        torch.cuda.set_device(self.dev)
        self.tracker_cfg = tracker_cfg or dict(      # tracklab/configs/modules/track/bpbreid_strong_sort.yaml:3-21
            ema_alpha=0.9, mc_lambda=0.995, max_dist=0.5, motion_criterium="iou", max_iou_distance=0.8, max_oks_distance=0.7,
            max_age=300, n_init=0, nn_budget=100, min_bbox_confidence=0.0, only_position_for_kf_gating=False,
            max_kalman_prediction_without_update=7, matching_strategy="strong_sort_matching", gating_thres_factor=1,
            w_kfgd=1, w_reid=1, w_st=1)
        if tracker == "strong_sort" and tracker_cfg is None:       # StrongSORT.__init__ defaults (strong_sort.py:19-32) + wrapper filter
            self.tracker_cfg = dict(max_dist=0.2, max_iou_dist=0.7, max_age=70, max_unmatched_preds=7, n_init=3, nn_budget=100,
                                    mc_lambda=0.995, ema_alpha=0.9)
        if tracker == "bot_sort" and tracker_cfg is None:          # BoTSORT.__init__ defaults (bot_sort.py:236-249), cmc_method none
            self.tracker_cfg = dict(track_high_thresh=0.45, new_track_thresh=0.6, track_buffer=30, match_thresh=0.8, proximity_thresh=0.5,
                                    appearance_thresh=0.25, frame_rate=30, lambda_=0.985)
        if tracker == "deep_oc_sort" and tracker_cfg is None:      # configs/modules/track/deep_oc_sort.yaml with cmc_off
            self.tracker_cfg = dict(det_thresh=0, max_age=50, min_hits=1, iou_threshold=0.22136877277096445, delta_t=1, asso_func="giou",
                                    inertia=0.3941737016672115, w_association_emb=0.75, alpha_fixed_emb=0.95, aw_param=0.5,
                                    embedding_off=False, cmc_off=True, aw_off=False, new_kf_off=False)
        self.pose = None
        if pose:
            from .backbones.rtmpose import rtmpose
            self.tracker_cfg = dict(self.tracker_cfg, motion_criterium="oks")
            self.pose = rtmpose(pose, device=self.dev, dtype=dtype, channels_last=True)
        self.model = yolox(detector, 1, device=self.dev, dtype=dtype, channels_last=True)
        # detector_split_precision (r06; dtype float32): the detector's convolutions in split mode too.  Its planes are unscaled:
        # range = float16's, and a non-finite prediction is ORed into the same flag the embeddings use (fails loudly in synchronize())
        self.det_split = bool(detector_split_precision) and dtype == torch.float32
        self.reid_arch = reid_arch
        # reid_split_precision (dtype float32, ResNet-50): the ReID backbone's convolutions in split mode -- fp32 values as (hi, lo) f16 pairs on the
        # 16-bit MFMA, fp32-class results (csrc/tlk_conv16.hip)
        self.reid = part_based_reid(parts, dim, device=self.dev, dtype=dtype, channels_last=True, arch=reid_arch,
                                    split_precision=reid_split_precision and dtype == torch.float32)
        # track capacity per stream: None = the tracker's own default.  Capacity is an allocation size in every bank (r04; <= 16384): the per-frame
        # lists stay in LDS while the scene is small.  4096 for BPBReID-StrongSORT, 2048 for BoT-SORT / Deep-OC-SORT, 1024 for plain StrongSORT
        # (its gallery is max_tracks x nn_budget x dim floats per stream).  An explicit value is handed to the bank AS IS: a bank that cannot
        # hold it raises TLK_ECAPACITY at creation (r03 clamped it silently, VERDICT r03 "what's missing" 1)
        if max_tracks is None:
            max_tracks = {"strong_sort": 1024, "bot_sort": 2048, "deep_oc_sort": 2048}.get(tracker, 4096)
        if tracker == "strong_sort":
            self.bank = _lib.SsortBank(dim, **self.tracker_cfg, min_confidence=0.4, wrapper_mode=True, img_w=width, img_h=height,
                                       n_streams=n_streams, device=device, max_tracks=max_tracks, max_dets=max_dets)
            self.row_dtype = _lib.SSORT_ROW
        elif tracker == "bot_sort":
            if camera_motion:
                self.tracker_cfg = dict(self.tracker_cfg, cmc_method="sparseOptFlow")
            self.bank = _lib.BoTSORTBank(dim, **self.tracker_cfg, min_confidence=0.4, wrapper_mode=True, n_streams=n_streams, device=device,
                                         max_tracks=max_tracks, max_dets=max_dets)
            self.row_dtype = _lib.BOTSORT_ROW
        elif tracker == "deep_oc_sort":
            self.bank = _lib.DeepOCSortBank(dim, **self.tracker_cfg, min_confidence=0.4, wrapper_mode=True, n_streams=n_streams, device=device,
                                            max_tracks=max_tracks, max_dets=max_dets)
            self.row_dtype = _lib.DEEPOCSORT_ROW
        else:
            self.bank = _lib.BpbssBank(parts, dim, **self.tracker_cfg, wrapper_mode=True, n_streams=n_streams, device=device,
                                       max_tracks=max_tracks, max_dets=max_dets)
            self.row_dtype = _lib.BPBSS_ROW
        B = n_streams * frames_per_step
        self.B = B
        dev = self.dev
        self.cmc = None
        if camera_motion:
            if tracker != "bot_sort":
                raise NotImplementedError("DetReidTrackPipeline(camera_motion=True) is wired for tracker='bot_sort' (sparse optical flow); plain StrongSORT's ECC "
                                          "runs in the module (wrappers.HipStrongSORT, ecc: true)")
            self.cmc = [_lib.CmcEstimator(height, width, downscale=2, device=device) for _ in range(n_streams)]
            self.cmc_stream = torch.cuda.Stream(device=dev)
        self.lb = torch.empty((B, size // 2, size // 2, 12), dtype=dtype, device=dev)
        self.crops = torch.zeros((B * max_dets, reid_hw[0], reid_hw[1], 3), dtype=dtype, device=dev)      # padding slots stay as they are
        self.det = {"ltwh": torch.zeros((B, max_dets, 4), dtype=torch.float32, device=dev),
                    "xyxy": torch.zeros((B, max_dets, 4), dtype=torch.float32, device=dev),
                    "scores": torch.zeros((B, max_dets), dtype=torch.float32, device=dev),
                    "cls": torch.zeros((B, max_dets), dtype=torch.int32, device=dev),
                    "counts": torch.zeros((B,), dtype=torch.int32, device=dev)}
        self.conf = torch.ones((B, max_dets), dtype=torch.float64, device=dev)        # RTMLibDetector: bbox_conf = 1.0
        if self.pose is not None:
            self.pose_hw = (256, 192)
            self.pose_crops = torch.empty((B * max_dets, 256, 192, 3), dtype=dtype, device=dev)
            self.pose_meta = torch.zeros((B * max_dets, 10), dtype=torch.float64, device=dev)
            self.xyxy32 = torch.zeros((B, max_dets, 4), dtype=torch.float32, device=dev)
            self.xyxy64 = torch.zeros((B, max_dets, 4), dtype=torch.float64, device=dev)
            self.pose_out = {"kps_xyc": torch.zeros((B * max_dets, 17, 3), dtype=torch.float64, device=dev),
                             "scores": torch.zeros((B * max_dets, 17), dtype=torch.float32, device=dev),
                             "conf": torch.zeros((B * max_dets,), dtype=torch.float32, device=dev)}
        self.id_off = torch.arange(B * max_dets, dtype=torch.int64, device=dev).reshape(B, max_dets)
        # r05: the ReID batch is DENSE -- only the real crops of the step are cropped, convolved and pooled (the reference's ReID wrapper batches real
        # detections only, wrappers/reid/kpreid_api.py:147-182): crop i of frame b sits at slot_base[b] + i, the libtlk convolutions read the live
        # crop count from n_live when they run (so one captured hipGraph serves every step), and the embeddings are gathered back into the
        # tracker's (frame, detection) layout through slot_of.  TLK_DENSE_REID=0 keeps the r04 slot layout (every slot convolved).
        self.dense_reid = (not self.global_feat) and __import__("os").environ.get("TLK_DENSE_REID", "1") != "0"
        self.slot_base = torch.zeros(B, dtype=torch.int32, device=dev)
        self.n_live = torch.full((1,), B * max_dets, dtype=torch.int32, device=dev)
        self.slot_of = torch.arange(B * max_dets, dtype=torch.int64, device=dev)
        # r05, overlap_stages: stage A (letterbox, detector, decode + NMS, crops) of step t + 1 runs on its own stream beside stage B (ReID
        # forward, hand-off) of step t; what A writes and B reads -- crops, slot bases, live count -- exists twice, B's results go to the
        # per-step ring `bufs` as before.  Only the plain BPBReID chain (no pose stage, no camera motion, dense batch) is wired for it.
        # r06: `None` = AUTO for the online shapes -- at most two frames per step in all (with 16-bit backbones a launch then has fewer tiles than the
        # chip has CUs and the two stages can fill each other's gaps) -- and AUTO means MEASURED: whether two HIP streams of
        # one process really run side by side is decided by how HIP dealt them to hardware queues and the queues to the command processor's
        # pipes, which the pipeline cannot choose and which shifts with every stream created before (and with an RCCL communicator in the
        # process): the same code measured 1.39x the serial pipeline, 0.99x, 0.69x and 0.50x in ONE process as other streams came and went
        # (profiles/r06_overlap_autotune.md; a spin-kernel probe of the two streams says "concurrent" in all four cases).  So the first
        # step() runs both modes on its own inputs (a few steps each, back to back, results discarded, tracker reset) and keeps the faster
        # one: never slower than serial.  overlap_stages=True / False (or TLK_PIPE_OVERLAP=1 / 0) forces a mode without the trial.  No trial for the
        # throughput shapes (24 frames per step fill the chip; their association hides on its side stream); with fp32 backbones the trial
        # measures the overlap slower (one-frame fp32 launches fill the chip: 46 vs 61 frames/s) and drops it.
        env_ov = __import__("os").environ.get("TLK_PIPE_OVERLAP", "auto")
        eligible = self.dense_reid and self.pose is None and not camera_motion
        if overlap_stages is None:
            mode = {"1": "on", "0": "off"}.get(env_ov, "trial" if n_streams * frames_per_step <= 2 else "off")
        elif overlap_stages == "auto":                    # the trial whatever the shape (bench legs)
            mode = "trial"
        else:
            mode = "on" if overlap_stages else "off"
        self._ov_mode = mode if eligible else "off"       # "on" / "off" / "trial" (decided by the first step)
        self.overlap = self._ov_mode == "on"
        self.trk_inline = False
        self.overlap_note = {"on": "on (forced)", "off": "off", "trial": "auto: not decided yet (first step)"}[self._ov_mode]
        self.overlap_trial = None
        self.sets = [{"crops": self.crops, "slot_base": self.slot_base, "n_live": self.n_live, "slot_of": self.slot_of,
                      "a_done": torch.cuda.Event(), "b_done": torch.cuda.Event()}]
        if self._ov_mode != "off":
            self.sets.append({"crops": torch.zeros_like(self.crops), "slot_base": torch.zeros_like(self.slot_base), "n_live": self.n_live.clone(),
                              "slot_of": self.slot_of.clone(), "a_done": torch.cuda.Event(), "b_done": torch.cuda.Event()})
            # the two stages only overlap when their streams sit on DIFFERENT hardware queues; HIP deals streams of one priority to a small
            # pool of queues round-robin, so two normal-priority streams may share one (measured: 219 frames/s then, 330 when they do not).
            # Streams of different priorities never share a queue: the detector stage gets the high-priority one (TLK_DET_PRIO overrides).
            # r06: each measured to run beside the compute stream and beside each other (pick_stream); the association stream below beside all three
            cur_ = torch.cuda.current_stream(dev)
            self.det_stream = pick_stream(dev, [cur_], priority=int(__import__("os").environ.get("TLK_DET_PRIO", "-1")))
            self.reid_stream = pick_stream(dev, [cur_, self.det_stream])
        # f16 and split-precision backbones carry activations as float16 (pairs): |x| > 65504 saturates to infinity.  The envelope is stated in
        # DESIGN.md (tests/test_gpu_precision.py measures it); past it the run must fail LOUDLY, not track on infinities: every step ORs
        # "an embedding is not finite" into a device flag that travels to pinned memory with the results and is checked in synchronize()
        self.check_finite = dtype != torch.float32 or bool(reid_split_precision) or bool(detector_split_precision)
        self.nf_flag = torch.zeros(2, dtype=torch.bool, device=dev)          # [0] embeddings (ReID stage), [1] detector predictions (split-mode detector): one writer each
        self.h_nf_flag = torch.zeros(2, dtype=torch.bool).pin_memory()
        self.nbuf = 2
        self.bufs = []
        row_bytes = self.row_dtype.itemsize
        for _ in range(self.nbuf):
            self.bufs.append({
                "ids": torch.zeros((B, max_dets), dtype=torch.int64, device=dev),
                "ltwh": torch.zeros((B, max_dets, 4), dtype=torch.float64, device=dev),
                "emb": torch.zeros((B, max_dets, parts, dim), dtype=torch.float32, device=dev),
                "vis": torch.zeros((B, max_dets, parts), dtype=torch.uint8, device=dev),
                "counts": torch.zeros((B,), dtype=torch.int32, device=dev),
                "trk_in": torch.zeros((B, max_dets, 7), dtype=torch.float64, device=dev) if self.global_feat else None,
                "kps": torch.zeros((B, max_dets, 17, 3), dtype=torch.float64, device=dev) if self.pose is not None else None,
                "rows": torch.zeros((B, max_dets, row_bytes), dtype=torch.uint8, device=dev),
                "ocnt": torch.zeros((B,), dtype=torch.int32, device=dev),
                "h_rows": torch.zeros((B, max_dets, row_bytes), dtype=torch.uint8).pin_memory(),
                "h_ocnt": torch.zeros((B,), dtype=torch.int32).pin_memory(),
                "h_ltwh": torch.zeros((B, max_dets, 4), dtype=torch.float64).pin_memory(),
                "h_dcnt": torch.zeros((B,), dtype=torch.int32).pin_memory(),
                "ready": torch.cuda.Event(), "done": torch.cuda.Event()})
        if self.cmc is not None:
            for b_ in self.bufs:
                b_["warps"] = torch.zeros((n_streams, frames_per_step, 6), dtype=torch.float64, device=dev)
                b_["cmc_done"] = torch.cuda.Event()
                b_["gate"] = torch.zeros((B,), dtype=torch.int32, device=dev)
        self.trk_stream = pick_stream(dev, [torch.cuda.current_stream(dev), getattr(self, "det_stream", None), getattr(self, "reid_stream", None)],
                                      priority=int(__import__("os").environ.get("TLK_TRK_PRIO", "0")))
        self.use_graph = use_graph
        self.det_graphs, self.reid_graph = {}, None
        self.step_idx = 0
        self.frames_done = 0
        self.ratio = min(size / height, size / width)
        self.kernel_events = []
        self.null_events = []
        self.record_kernel_events = False

    def reset(self, keep_ids: bool = False):
        """keep_ids: ByteTrack / BoT-SORT keep their id counter (the reference's class-level BaseTrack._count); the other trackers' counters are
        re-created with the tracker in the reference, so they restart at 1 either way."""
        torch.cuda.synchronize(self.dev)
        if keep_ids and self.tracker in ("byte_track", "bot_sort"):
            self.bank.reset(-1, keep_ids=True)
        else:
            self.bank.reset(-1)
        for est in (getattr(self, "cmc", None) or []):
            est.reset()
        self.frames_done = 0
        self.nf_flag.zero_()
        self.h_nf_flag.zero_()

    def synchronize(self):
        if self._ov_mode != "off":
            self.det_stream.synchronize()
            self.reid_stream.synchronize()
        self.trk_stream.synchronize()
        torch.cuda.current_stream(self.dev).synchronize()
        if self.check_finite and bool(self.h_nf_flag.any()):
            which = "detector predictions" if bool(self.h_nf_flag[1]) and not bool(self.h_nf_flag[0]) else "ReID embeddings"
            self.nf_flag.zero_()
            self.h_nf_flag.zero_()
            raise _lib.TlkError(f"{which} are not finite: the float16 / split-precision backbones saturated (an activation beyond +-65504, "
                                "float16's range; the split-precision ReID network follows larger activations with its plane scales, r06); this network "
                                "needs dtype float32 (the reference's precision) -- DESIGN.md, precision envelope")

    def _feat_probe(self):
        """a zero-size stand-in with the dtype / layout of the ReID feature map (what fused_head_ok looks at)"""
        p = self.__dict__.get("_fprobe")
        if p is None:
            p = torch.empty((1, self.D, 1, 1), dtype=self.dtype, device=self.dev).contiguous(memory_format=torch.channels_last)
            if getattr(self.reid, "split_precision", False):
                p = p.float()
            self._fprobe = p
        return p

    def _graphed(self, cache, key, fn):
        ent = cache.get(key)
        if ent is None:
            side = torch.cuda.Stream(device=self.dev)
            side.wait_stream(torch.cuda.current_stream(self.dev))
            with torch.cuda.stream(side):
                for _ in range(2):
                    fn()
            torch.cuda.current_stream(self.dev).wait_stream(side)
            torch.cuda.synchronize(self.dev)
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                out = fn()
            ent = (g, out)
            cache[key] = ent
        ent[0].replay()
        return ent[1]

    def _autotune_overlap(self, frames, synth_head, warm: int = 5, timed: int = 14):
        """AUTO mode, first step: the stream arrangements on this step's inputs, back to back, results discarded; keep the fastest (see __init__).
        Three candidates -- (a) serial stages, association on its side stream (the r01-r05 arrangement); (b) stages overlapped on two streams;
        (c) everything on ONE stream, association inline.  (c) is in the list because (a) is not immune either: its association stream is a
        second hardware queue, and with an unlucky assignment the SERIAL pipeline itself was measured at 114 instead of 251 frames/s
        (tests/test_gpu_zz_stage_overlap.py, one idle stream created before it); inline costs the association's ~0.2 ms per frame and
        cannot be hit by queue placement."""
        import time
        cands = [("serial stages, association on a side stream", False, False), ("stages overlapped, association on a side stream", True, False),
                 ("one stream (association inline)", False, True)]
        rates = []
        self._ov_mode = "tuning"
        for _, ov, inline in cands:
            self.overlap, self.trk_inline = ov, inline
            for _ in range(warm):
                self._step(frames, synth_head, False, None)
            torch.cuda.synchronize(self.dev)
            t0 = time.perf_counter()
            for _ in range(timed):
                self._step(frames, synth_head, False, None)
            torch.cuda.synchronize(self.dev)
            rates.append(timed / (time.perf_counter() - t0))
        best = 0                                   # the r05 arrangement unless another one is clearly (5 %) faster
        for i in (1, 2):
            if rates[i] > 1.05 * rates[best]:
                best = i
        self.overlap, self.trk_inline = cands[best][1], cands[best][2]
        self._ov_mode = "on" if self.overlap else "off_tuned"
        self.overlap_trial = {"serial_steps_per_s": rates[0], "overlapped_steps_per_s": rates[1], "one_stream_steps_per_s": rates[2], "steps_timed": timed,
                              "chosen": cands[best][0]}
        self.overlap_note = (f"auto: measured {rates[0] * self.B:.1f} / {rates[1] * self.B:.1f} / {rates[2] * self.B:.1f} frames/s (serial stages / stages overlapped / one "
                             f"stream) over {timed} back-to-back steps of the first step's inputs -> {cands[best][0]}")
        self.reset()                              # the trial's tracks, ids and flags are gone: the real first step starts from a fresh tracker
        self.step_idx = 0

    @torch.no_grad()
    def step(self, frames: torch.Tensor, synth_head: torch.Tensor | None = None, fetch: bool = True, sink=None):
        """frames (S*F, H, W, 3) uint8 RGB on device (channel contract: module docstring); sink / ``self.frames_free`` as in
        ``DetTrackPipeline.step``."""
        if self._ov_mode == "trial":
            self._autotune_overlap(frames, synth_head)
        return self._step(frames, synth_head, fetch, sink)

    @torch.no_grad()
    def _step(self, frames, synth_head, fetch, sink):
        S, F, maxd = self.S, self.F, self.maxd
        buf = self.bufs[self.step_idx % self.nbuf]
        st = self.sets[self.step_idx % len(self.sets)] if self.overlap else self.sets[0]
        self.step_idx += 1
        main = torch.cuda.current_stream(self.dev)
        if self.overlap:
            entry = torch.cuda.Event()
            entry.record(main)                      # the caller's frames are ready at this point of ITS stream (which carries none of our work)
            sa, sb = self.det_stream, self.reid_stream
            sa.wait_event(entry)
            sa.wait_event(buf["done"])              # the tracker is done with this ring entry (two steps ago)
            sa.wait_event(st["b_done"])             # the ReID stage of two steps ago has read this set of crops
        else:
            sa = sb = main
            main.wait_event(buf["done"])

        # ---- stage A: letterbox + detector, decode + NMS, crops (everything that reads `frames`)
        with torch.cuda.stream(sa):
            def det_fwd():
                x, _ = _lib.letterbox(frames, self.size, "focus_nhwc", self.dtype, out=self.lb, swap_rb=True)
                return self.model(x, focused=True, split=self.det_split)
            pred = self._graphed(self.det_graphs, frames.data_ptr(), det_fwd) if self.use_graph else det_fwd()
            if self.det_split:
                self.nf_flag[1:2].logical_or_(torch.logical_not(torch.isfinite(pred).all()))
            if synth_head is not None:
                pred = torch.add(synth_head, torch.nan_to_num(pred), alpha=0.0)
            _lib.yolox_decode_nms(pred, self.size, float(np.float32(self.ratio)), self.W, self.H, maxd, self.nms_thr,
                                  self.score_thr, out=self.det, trk_in=buf["trk_in"], det_id_base=self.frames_done * maxd, category_id=1.0)
            if self.record_kernel_events:
                self._record_null_pair()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
            if self.global_feat:                    # StrongSORT._get_features: int-truncated boxes, Pillow resize, ImageNet normalisation
                crops = _lib.roi_crop_pil_resize_norm(frames, buf["trk_in"], self.det["counts"], self.reid_hw[0], self.reid_hw[1],
                                                      "nhwc", self.dtype, out=st["crops"])
            else:
                if self.dense_reid:
                    _lib.crop_slot_bases(self.det["counts"], maxd, st["slot_base"], st["n_live"], st["slot_of"])
                crops = _lib.roi_crop_resize_norm(frames, self.det["ltwh"], self.det["counts"], self.reid_hw[0], self.reid_hw[1],
                                                  "nhwc", self.dtype, out=st["crops"], slot_base=st["slot_base"] if self.dense_reid else None)
            if self.record_kernel_events:
                e1.record()
                self.kernel_events.append((e0, e1))
            if self.pose is not None:
                # pose stage (rtmlib RTMPose(image, bboxes)): affine crops of every box -> network -> SimCC decode, all in HBM
                # boxes as RTMPose.process sees them: detections.bbox.ltrb() of the SANITIZED float32 bbox_ltwh the detector stored
                # (l, t, l + w, t + h in float32, widened), not the decode kernel's unclipped xyxy
                ltwh32 = self.det["ltwh"]
                self.xyxy32[..., :2].copy_(ltwh32[..., :2])
                torch.add(ltwh32[..., :2], ltwh32[..., 2:], out=self.xyxy32[..., 2:])
                self.xyxy64.copy_(self.xyxy32)
                pcrops, _ = _lib.pose_crop_warp_norm(frames, self.xyxy64, self.det["counts"], 192, 256, "nhwc", self.dtype,
                                                     out=self.pose_crops, meta=self.pose_meta, swap_rb=True)
            self.frames_free = torch.cuda.Event()       # the crop kernels were the last readers of `frames` ...
            if self.cmc is not None:
                # ... unless the camera-motion estimators read them too: frame by frame per stream, on their own stream (about 25 small launches per frame
                # that would otherwise sit on the main stream between the backbone launches)
                buf["gate"].copy_(self.det["counts"])       # (this step's own copy: the next step's decode overwrites det["counts"] while the estimators may still run)
                fed = torch.cuda.Event()
                fed.record(sa)
                with torch.cuda.stream(self.cmc_stream):
                    self.cmc_stream.wait_event(fed)
                    sp = C.c_void_p(self.cmc_stream.cuda_stream)
                    for s_ in range(S):
                        for f_ in range(F):
                            # gated on the frame's detection count, on the device: the reference skips GMC.apply for a frame without detections
                            self.cmc[s_].apply_dev(frames[s_ * F + f_], stream_ptr=sp, out=buf["warps"][s_, f_], count=buf["gate"][s_ * F + f_:s_ * F + f_ + 1])
                    buf["cmc_done"].record(self.cmc_stream)
                    self.frames_free.record(self.cmc_stream)
            else:
                self.frames_free.record(sa)
            # what the association needs of the detector's output leaves the shared `det` arrays here (the next step's decode overwrites them)
            buf["ltwh"].copy_(self.det["ltwh"])
            buf["counts"].copy_(self.det["counts"])
            st["a_done"].record(sa)

        # ---- stage B: pose / ReID forward, hand-off to the association stream
        with torch.cuda.stream(sb):
            if self.overlap:
                sb.wait_event(st["a_done"])
            if self.pose is not None:
                if self.use_graph:
                    sx, sy = self._graphed(self.__dict__.setdefault("_pg", {}), 0, lambda: self.pose(pcrops))
                else:
                    sx, sy = self.pose(pcrops)
                _lib.simcc_decode(sx, sy, self.pose_meta, 192, 256, 2.0, out=self.pose_out)
                buf["kps"].copy_(self.pose_out["kps_xyc"].view(self.B, maxd, 17, 3))
            if self.dense_reid:
                _lib.conv_set_dynamic_batch(st["n_live"])      # (a captured graph keeps the pointer: every replay reads the step's own count)
            # r06: the network's graph ends at the feature map; the part-based head is ONE libtlk launch behind it (tlk_reid_part_head) that pools the
            # step's LIVE crops only, writes their rows straight into the tracker's (frame, slot) hand-off layout (dense batch: through the slot
            # bases), zero-fills the padding slots and ORs "an embedding is not finite" into nf_flag -- it replaces the 6-channel library
            # convolution, softmax, bmm, division, amax, two index_select gathers and the isfinite passes (and ADVICE r05: the check no longer
            # sees padding rows, whose content used to be whatever a never-convolved row of the dense batch held)
            fused_head = (not self.global_feat) and self.reid.fused_head_ok(self._feat_probe())
            net = self.reid.features if fused_head else self.reid
            try:
                if self.use_graph:
                    res = self._graphed(self.__dict__.setdefault("_rg", {}), id(st), lambda: net(crops))
                else:
                    res = net(crops)
            finally:
                if self.dense_reid:
                    _lib.conv_set_dynamic_batch(None)
            if fused_head:
                self.reid.head(res, counts=buf["counts"], slot_base=st["slot_base"] if self.dense_reid else None, max_dets=maxd,
                               out_emb=buf["emb"], out_vis=buf["vis"], flag=self.nf_flag[0:1] if self.check_finite else None)
                if self.check_finite:
                    self.h_nf_flag.copy_(self.nf_flag, non_blocking=True)
            else:
                emb, vis = res
                # hand-off buffers for the association stream (detector ltwh is float32: widen like numpy would)
                if self.dense_reid:       # dense batch -> (frame, detection) slots; padding slots receive some valid row, the tracker reads counts[b] of them
                    torch.index_select(emb.reshape(self.B * maxd, self.K * self.D), 0, st["slot_of"], out=buf["emb"].view(self.B * maxd, self.K * self.D))
                    torch.index_select(vis.reshape(self.B * maxd, self.K).to(torch.uint8), 0, st["slot_of"], out=buf["vis"].view(self.B * maxd, self.K))
                else:
                    buf["emb"].copy_(emb.view(self.B, maxd, self.K, self.D))
                    buf["vis"].copy_(vis.view(self.B, maxd, self.K))
                if self.check_finite:
                    # live rows only: frame b's slots [0, counts[b]) (the padding slots of a dense batch repeat some row of it -- of a step without
                    # detections, a row no convolution wrote)
                    livem = torch.arange(maxd, device=self.dev)[None, :] < buf["counts"][:, None]
                    bad = torch.logical_not(torch.isfinite(buf["emb"]).flatten(2).all(-1)) & livem
                    self.nf_flag[0:1].logical_or_(bad.any())
                    self.h_nf_flag.copy_(self.nf_flag, non_blocking=True)
            torch.add(self.id_off, self.frames_done * maxd, out=buf["ids"])
            buf["ready"].record(sb)
            st["b_done"].record(sb)
        self.frames_done += S * F
        ts = sb if self.trk_inline else self.trk_stream      # (trk_inline: the association behind stage B on ITS stream -- no third queue; see _autotune_overlap)
        with torch.cuda.stream(ts):
            ts.wait_event(buf["ready"])
            if self.cmc is not None:
                ts.wait_event(buf["cmc_done"])
                self.bank.update_dev(buf["trk_in"].data_ptr(), buf["emb"].data_ptr(), buf["counts"].data_ptr(), F, buf["rows"].data_ptr(),
                                     maxd, buf["ocnt"].data_ptr(), C.c_void_p(ts.cuda_stream), warps=buf["warps"].data_ptr())
            elif self.global_feat:
                self.bank.update_dev(buf["trk_in"].data_ptr(), buf["emb"].data_ptr(), buf["counts"].data_ptr(), F, buf["rows"].data_ptr(),
                                     maxd, buf["ocnt"].data_ptr(), C.c_void_p(ts.cuda_stream))
            else:
                self.bank.update_dev(buf["ids"].data_ptr(), buf["ltwh"].data_ptr(), buf["emb"].data_ptr(), buf["vis"].data_ptr(),
                                     self.conf.data_ptr(), buf["counts"].data_ptr(), F, buf["rows"].data_ptr(), maxd,
                                     buf["ocnt"].data_ptr(), C.c_void_p(ts.cuda_stream),
                                     kps=buf["kps"].data_ptr() if self.pose is not None else None)
            if sink is not None:
                sink(self.result_tensors(buf))
            if fetch:       # every host-visible result of the step is in pinned memory before `done`
                buf["h_rows"].copy_(buf["rows"], non_blocking=True)
                buf["h_ocnt"].copy_(buf["ocnt"], non_blocking=True)
                buf["h_ltwh"].copy_(buf["ltwh"], non_blocking=True)
                buf["h_dcnt"].copy_(buf["counts"], non_blocking=True)
            buf["done"].record(ts)
        self.last = buf
        return (buf["h_rows"], buf["h_ocnt"]) if fetch else (buf["rows"], buf["ocnt"])

    def result_tensors(self, buf):
        return {"rows": buf["rows"], "ocnt": buf["ocnt"], "ltwh": buf["ltwh"], "dcnt": buf["counts"]}

    def host_results(self, buf):
        return {"rows": buf["h_rows"].numpy(), "ocnt": buf["h_ocnt"].numpy(), "ltwh": buf["h_ltwh"].numpy(), "dcnt": buf["h_dcnt"].numpy()}

    done_key = "done"

    def track_columns(self, rows, ocnt):
        """(frames, maxd, row bytes) uint8 row blocks + (frames,) counts -> flat (frame index, det_id, track_id, ltwh, conf) of the valid rows."""
        r = np.ascontiguousarray(rows).view(self.row_dtype).reshape(rows.shape[0], rows.shape[1])
        f, i = np.nonzero(np.arange(r.shape[1])[None, :] < ocnt[:, None])
        r = r[f, i]
        names = r.dtype.names
        if "kf_ltwh" in names:                   # BPBReID-StrongSORT rows
            tl, conf = r["kf_ltwh"].reshape(-1, 4), np.ones(len(r))
        else:                                    # plain StrongSORT / BoT-SORT / Deep-OC-SORT rows: ltrb + the tracker's confidence column
            b = r["ltrb"].reshape(-1, 4)
            tl = np.stack([b[:, 0], b[:, 1], b[:, 2] - b[:, 0], b[:, 3] - b[:, 1]], axis=1).reshape(-1, 4)
            conf = r["conf"] if "conf" in names else r["score"]
        return f, r["det_id"].astype(np.int64), r["track_id"].astype(np.float64), tl, np.asarray(conf, dtype=np.float64)

    def rows_numpy(self, h_rows, h_ocnt):
        """(S, F) nested lists of structured row arrays from the pinned result block."""
        r = h_rows.numpy().view(self.row_dtype).reshape(self.S, self.F, self.maxd)
        c = h_ocnt.numpy().reshape(self.S, self.F)
        return [[r[s, f, :max(int(c[s, f]), 0)].copy() for f in range(self.F)] for s in range(self.S)], c

    def close(self):
        self.det_graphs.clear()
        self.__dict__.pop("_rg", None)
        self.__dict__.pop("_pg", None)
        for est in (getattr(self, "cmc", None) or []):
            est.close()
        self.bank.close()
