"""-m gpu: the per-video online loop with the columnar detection table (tracklab_amd/engine.py) against the oracle chain run
frame by frame (C decode/NMS + C tracker), for both pipeline shapes."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _inputs(seed, n_obj, n_frames, ratio):
    from tracklab_amd.synth import SyntheticStream, render_frame, synth_yolox_head
    rng = np.random.default_rng(seed)
    heads, frames = [], []
    for fr in SyntheticStream(seed, n_obj, n_frames):
        heads.append(synth_yolox_head(rng, fr["dets"][:, :4], ratio=ratio))
        frames.append(render_frame(rng, fr["gt_boxes"]))
    return np.stack(heads), frames


def _detector_rows(orc, head, ratio):
    boxes, _, _ = orc.yolox_postprocess(head, 640, float(np.float32(ratio)))
    l = np.maximum(0, np.minimum(boxes[:, 0], 1918)).astype(np.float32); t = np.maximum(0, np.minimum(boxes[:, 1], 1078)).astype(np.float32)
    r = np.maximum(1, np.minimum(boxes[:, 2], 1919)).astype(np.float32); b = np.maximum(1, np.minimum(boxes[:, 3], 1079)).astype(np.float32)
    return np.stack([l, t, r - l, b - t], axis=1)


def test_video_engine_det_ocsort_table_matches_oracle_chain(orc):
    from tracklab_amd import gpu_pipeline as gp
    from tracklab_amd.engine import HipVideoEngine
    F, T = 4, 10                                            # 10 frames = 2 full steps + a partial one
    pipe = gp.DetTrackPipeline("s", n_streams=1, frames_per_step=F, max_dets=64, use_graph=False)
    heads, frames = _inputs(21, 12, T, pipe.ratio)
    eng = HipVideoEngine(pipe)
    df = eng.video_loop(iter(frames), video_id=5, synth_heads=lambda t0, n: heads[t0:t0 + n])
    trk = orc.OCSort(**pipe.tracker_cfg["hyper"])
    assert list(df.columns) == ["image_id", "video_id", "category_id", "bbox_ltwh", "bbox_conf", "track_id", "track_bbox_ltwh", "track_bbox_conf"]
    assert (df.video_id == 5).all() and (df.category_id == 1).all() and (df.bbox_conf == 1.0).all()
    for f in range(T):
        ltwh = _detector_rows(orc, heads[f], pipe.ratio)
        sub = df[df.image_id == f]
        np.testing.assert_allclose(np.stack(sub.bbox_ltwh.to_list()), ltwh, rtol=2e-6, atol=2e-4)     # expf: 1-2 ulp vs libm
        ltwh = np.stack(sub.bbox_ltwh.to_list())               # the tracker is checked on the boxes the table holds
        ids = f * 64 + np.arange(len(ltwh))
        np.testing.assert_array_equal(sub.index.to_numpy(), ids)
        dets = np.zeros((len(ltwh), 7))
        dets[:, 0], dets[:, 1] = ltwh[:, 0], ltwh[:, 1]
        dets[:, 2], dets[:, 3] = (ltwh[:, 0] + ltwh[:, 2]).astype(np.float32), (ltwh[:, 1] + ltwh[:, 3]).astype(np.float32)
        dets[:, 4], dets[:, 5], dets[:, 6] = 1.0, 1.0, ids
        exp = orc.ocsort_wrapper_step(trk, dets, pipe.tracker_cfg["min_confidence"])
        got = sub[sub.track_id.notna()]
        assert len(got) == len(exp)
        order = np.argsort(exp[:, 7])
        np.testing.assert_array_equal(got.index.to_numpy(), exp[order, 7].astype(int))
        np.testing.assert_array_equal(got.track_id.to_numpy(), exp[order, 4])
    second = eng.video_loop(np.stack(frames), video_id=6, synth_heads=lambda t0, n: heads[t0:t0 + n])      # reset between videos
    np.testing.assert_array_equal(second.track_id.to_numpy(), df.track_id.to_numpy())
    pipe.close()


def test_video_engine_det_reid_strongsort_runs_and_keeps_ids_stable(orc):
    from tracklab_amd import gpu_pipeline as gp
    from tracklab_amd.engine import HipVideoEngine
    F, T = 3, 7
    pipe = gp.DetReidTrackPipeline("s", n_streams=1, frames_per_step=F, max_dets=32, dim=64, use_graph=False)
    heads, frames = _inputs(22, 10, T, pipe.ratio)
    df = HipVideoEngine(pipe).video_loop(frames, synth_heads=lambda t0, n: heads[t0:t0 + n])
    per_frame = df.groupby("image_id").size()
    assert len(per_frame) == T and (per_frame >= 9).all()
    tracked = df[df.track_id.notna()]
    assert len(tracked) >= 0.9 * len(df)                        # n_init = 0: tracks are confirmed on their first frame
    assert tracked.track_id.nunique() <= 14                     # 10 objects, near-identical boxes frame to frame: ids persist
    assert np.isfinite(np.stack(tracked.track_bbox_ltwh.to_list())).all()
    pipe.close()


@pytest.mark.parametrize("tracker", ["strong_sort", "bot_sort", "deep_oc_sort", "bot_sort+cmc"])
def test_pipeline_global_feature_trackers_match_oracle_fed_with_gpu_embeddings(orc, tracker):
    """DetReidTrackPipeline.step with each tracker that owns a global-feature ReID net: the rows the bank produced on the device
    equal the C oracle's for the same detector rows and the embeddings the GPU network produced."""
    import torch
    from tracklab_amd import gpu_pipeline as gp
    F, T = 4, 12
    cmc = tracker.endswith("+cmc")           # the reference's default cmc_method sparseOptFlow, estimated on the device inside the fused step
    tracker = tracker.split("+")[0]
    pipe = gp.DetReidTrackPipeline("s", n_streams=1, frames_per_step=F, max_dets=32, use_graph=False, tracker=tracker, camera_motion=cmc)
    heads, frames = _inputs(33, 10, T, pipe.ratio)
    if cmc:
        # frames WITHOUT detections: the reference returns before GMC.apply (bot_sort_api.py:59-60), so the next warp spans the frames
        # either side of the gap; the fused step gates its estimator on the frame's detection count on the device (tlk_cmc_apply_dev_gated).
        # One in the middle of a step, two in a row across a step boundary
        for t in (2, 7, 8):
            heads[t][:, 4] = 0.0
    gmc = orc.SparseOptFlowGMC(1080, 1920, 2) if cmc else None
    cfg = pipe.tracker_cfg
    ref = {"strong_sort": lambda: orc.PlainStrongSORT(pipe.D, **cfg, img_w=1920, img_h=1080), "bot_sort": lambda: orc.BoTSORT(pipe.D, **{k: v for k, v in cfg.items() if k != "cmc_method"}),
           "deep_oc_sort": lambda: orc.DeepOCSort(pipe.D, **cfg)}[tracker]()
    n_rows = 0
    for k in range(T // F):
        fr = torch.from_numpy(np.stack(frames[k * F:(k + 1) * F])).cuda()
        h_rows, h_cnt = pipe.step(fr, torch.from_numpy(heads[k * F:(k + 1) * F]).cuda())
        pipe.synchronize()
        rows, cnt = pipe.rows_numpy(h_rows, h_cnt)
        emb = pipe.last["emb"].cpu().numpy().reshape(1, F, pipe.maxd, pipe.D)
        trk_in = pipe.last["trk_in"].cpu().numpy().reshape(1, F, pipe.maxd, 7)
        dcnt = pipe.last["counts"].cpu().numpy()
        for f in range(F):
            n = int(dcnt[f])
            if cmc and n:
                exp = ref.update(trk_in[0, f, :n], emb[0, f, :n], warp=gmc.apply(frames[k * F + f]))
            else:                                                     # a frame without detections reaches neither the estimator nor the tracker
                exp = ref.update(trk_in[0, f, :n], emb[0, f, :n]) if n else np.zeros((0, 8))
            if cmc:
                assert (n == 0) == (k * F + f in (2, 7, 8))
            got = rows[0][f]
            assert len(got) == len(exp), (tracker, k, f)
            np.testing.assert_array_equal(got["det_id"].astype(np.int64), exp[:, 7].astype(np.int64))
            np.testing.assert_array_equal(got["track_id"].astype(np.int64), exp[:, 4].astype(np.int64))
            # (with camera motion: the two estimators agree to ~1e-12 in the warp -- the refit's summation order -- times up to 1920 px)
            np.testing.assert_allclose(got["ltrb"], exp[:, :4], rtol=1e-9, atol=1e-6 if cmc else 1e-7)
            n_rows += len(exp)
    assert n_rows > 40
    pipe.close()


@pytest.mark.parametrize("tracker", ["strong_sort", "bot_sort", "deep_oc_sort"])
def test_video_engine_runs_the_global_feature_trackers(tracker):
    """The per-video loop fills the table for every ReID tracker's row type (ltrb + the tracker's own confidence column)."""
    from tracklab_amd import gpu_pipeline as gp
    from tracklab_amd.engine import HipVideoEngine
    F, T = 3, 8
    pipe = gp.DetReidTrackPipeline("s", n_streams=1, frames_per_step=F, max_dets=32, use_graph=False, tracker=tracker)
    heads, frames = _inputs(24, 10, T, pipe.ratio)
    df = HipVideoEngine(pipe).video_loop(frames, synth_heads=lambda t0, n: heads[t0:t0 + n])
    assert df.image_id.nunique() == T
    tracked = df[df.track_id.notna()]
    first = 0 if tracker != "strong_sort" else 2                # n_init = 3: plain StrongSORT reports from its third hit on
    late = tracked[tracked.image_id >= first]
    assert len(late) >= 0.8 * len(df[df.image_id >= first]) and late.track_id.nunique() <= 14
    ltwh = np.stack(late.track_bbox_ltwh.to_list())
    assert np.isfinite(ltwh).all() and (ltwh[:, 2:] > 0).all() and np.isfinite(late.track_bbox_conf.to_numpy()).all()
    pipe.close()


class _Recorder:
    def __init__(self, per_image):
        self.events = []
        if per_image:
            self.on_image_loop_end = lambda engine, image_metadata, image, image_idx, detections: self.events.append(("image", int(image_idx), len(detections)))

    def on_dataset_track_start(self, engine): self.events.append(("dataset_start",))
    def on_dataset_track_end(self, engine): self.events.append(("dataset_end",))
    def on_video_loop_start(self, engine, video_metadata, video_idx, index): self.events.append(("video_start", int(video_idx)))
    def on_video_loop_end(self, engine, video_metadata, video_idx, detections, image_pred): self.events.append(("video_end", int(video_idx), len(detections)))
    def on_module_step_start(self, engine, task, batch): self.events.append(("step_start", task))
    def on_module_step_end(self, engine, task, batch, detections): self.events.append(("step_end", task, len(detections)))


class _ProgressLike:
    """The hooks of tracklab.callbacks.Progressbar with its signatures and its bookkeeping (callbacks/progress.py:40-78): the per-task bar is
    created in on_module_start from engine.models[task] / len(dataloader) and indexed in on_module_step_end / on_module_end -- an engine that
    emits step hooks without on_module_start raises KeyError there."""

    def __init__(self):
        self.task_pbars, self.closed, self.video_id = {}, [], None

    def on_video_loop_start(self, engine, video_metadata, video_idx, index):
        self.video_id = video_idx

    def on_module_start(self, engine, task, dataloader):
        if hasattr(engine.models[task], "process_video"):
            length = len(engine.img_metadatas[engine.img_metadatas.video_id == self.video_id])
        else:
            length = len(dataloader)
        self.task_pbars[task] = {"total": length, "n": 0}

    def on_module_step_end(self, engine, task, batch, detections):
        self.task_pbars[task]["n"] += 1

    def on_module_end(self, engine, task, detections):
        bar = self.task_pbars[task]
        assert bar["n"] == bar["total"], bar
        self.closed.append(task)


def test_tracking_engine_fires_the_callback_hooks_and_online_equals_resident():
    """HipTrackingEngine.track_dataset over two videos (engine/engine.py:105-126): hook order, image ids mapped back to the dataset's,
    and the `online` drain (per-image callbacks, engine/video.py:93-117) produces the same table as the HBM-resident mode."""
    import pandas as pd
    from types import SimpleNamespace
    from tracklab_amd import gpu_pipeline as gp
    from tracklab_amd.engine import HipTrackingEngine
    F, T = 4, 10
    pipe = gp.DetTrackPipeline("s", n_streams=1, frames_per_step=F, max_dets=64, use_graph=False)
    streams = {7: _inputs(41, 10, T, pipe.ratio), 9: _inputs(42, 8, T, pipe.ratio)}
    imgs = pd.DataFrame({"video_id": [7] * T + [9] * T, "frame": list(range(T)) * 2, "file_path": [f"{v}/{f}" for v in (7, 9) for f in range(T)]},
                        index=pd.Index(list(range(700, 700 + T)) + list(range(900, 900 + T)), name="id"))
    videos = pd.DataFrame({"name": ["a", "b"]}, index=pd.Index([7, 9], name="id"))
    state = SimpleNamespace(image_metadatas=imgs, video_metadatas=videos)
    load = lambda p: streams[int(p.split("/")[0])][1][int(p.split("/")[1])]               # noqa: E731
    heads = lambda vid, t0, n: streams[vid][0][t0:t0 + n]                                # noqa: E731
    tables = {}
    for per_image in (False, True):
        rec = _Recorder(per_image)
        got = {}
        rec_end = rec.on_video_loop_end
        rec.on_video_loop_end = lambda engine, video_metadata, video_idx, detections, image_pred: (got.__setitem__(int(video_idx), detections), rec_end(engine, video_metadata, video_idx, detections, image_pred))
        bar = _ProgressLike()
        eng = HipTrackingEngine(modules=[], tracker_state=state, num_workers=0, callbacks={"rec": rec, "progress": bar}, pipeline=pipe, image_loader=load,
                                synth_heads=heads)
        eng.track_dataset()
        assert bar.closed == ["hip_fused_pipeline"] * 2                              # one module start / end pair per video, every bar complete
        ev = rec.events
        assert ev[0] == ("dataset_start",) and ev[-1] == ("dataset_end",)
        assert [e for e in ev if e[0] in ("video_start", "video_end")] == [("video_start", 7), ("video_end", 7, len(got[7])), ("video_start", 9), ("video_end", 9, len(got[9]))]
        assert sum(e[0] == "step_start" for e in ev) == 2 * 3                        # ceil(10 / 4) fused steps per video
        if per_image:
            assert [e[1] for e in ev if e[0] == "image"] == list(imgs.index)            # every image, in order, ids of the dataset
            assert sum(e[2] for e in ev if e[0] == "image") == len(got[7]) + len(got[9])
            assert sum(e[0] == "step_end" for e in ev) == 6
        else:
            assert sum(e[0] == "step_end" for e in ev) == 2                          # resident: one table per video
        for v in (7, 9):
            assert set(got[v].image_id) == set(imgs.index[imgs.video_id == v]) and got[v].track_id.notna().sum() > 40
        assert not set(got[7].index) & set(got[9].index)                             # detection ids are dataset-global
        assert got[9].index.min() >= 3 * F * 64
        tables[per_image] = got
    for v in (7, 9):
        a, b = tables[False][v], tables[True][v]
        np.testing.assert_array_equal(a.index.to_numpy(), b.index.to_numpy())
        np.testing.assert_array_equal(a.image_id.to_numpy(), b.image_id.to_numpy())
        np.testing.assert_array_equal(a.track_id.to_numpy(), b.track_id.to_numpy())
        np.testing.assert_array_equal(np.stack(a.bbox_ltwh.to_list()), np.stack(b.bbox_ltwh.to_list()))
    pipe.close()


@pytest.mark.parametrize("kind", ["oc_sort", "byte_track", "bpbreid", "strong_sort"])
def test_evaluators_fed_from_the_hbm_resident_table_equal_the_host_fed_ones(kind):
    """HOTA and the CLEAR-MOT / ID counts of a video computed from the per-video table WHERE THE ENGINE LEFT IT (engine.DeviceStepLog in
    HBM; evaluate.evaluate_device_log -> tlk_hota_sequence_dev_f64 / tlk_clear_sequence_dev_f64) equal the same evaluators fed with the fetched
    table through host arrays, and the host (numpy) evaluators: every count exactly, HOTA's floating-point sums to 1e-12. A partial last step,
    frames without detections, misses and churn are in the stream."""
    from tracklab_amd import evaluate, gpu_pipeline as gp
    from tracklab_amd.engine import HipVideoEngine
    from tracklab_amd.synth import SyntheticStream, render_frame, synth_yolox_head
    F, T = 4, 22
    if kind in ("oc_sort", "byte_track"):
        pipe = gp.DetTrackPipeline("s", n_streams=1, frames_per_step=F, max_dets=64, use_graph=False, tracker=kind)
    elif kind == "bpbreid":
        pipe = gp.DetReidTrackPipeline("s", n_streams=1, frames_per_step=F, max_dets=32, dim=64, use_graph=False)
    else:
        pipe = gp.DetReidTrackPipeline("s", n_streams=1, frames_per_step=F, max_dets=32, use_graph=False, tracker=kind)
    rng = np.random.default_rng(41)
    heads, frames, gt = [], [], {"frame": [], "track_id": [], "ltwh": []}
    for t, fr in enumerate(SyntheticStream(41, 14, T, miss_prob=0.15, churn_period=6)):
        heads.append(synth_yolox_head(rng, fr["dets"][:, :4], ratio=pipe.ratio))
        frames.append(render_frame(rng, fr["gt_boxes"]))
        b = np.asarray(fr["gt_boxes"], dtype=np.float64)
        gt["frame"].extend([t + 1] * len(b)); gt["track_id"].extend(int(i) for i in fr["gt_all_ids"])
        gt["ltwh"].extend(np.column_stack([b[:, 0], b[:, 1], b[:, 2] - b[:, 0], b[:, 3] - b[:, 1]]).tolist())
    heads = np.stack(heads)
    heads[9][:, 4] = 0.0; heads[10][:, 4] = 0.0                # two frames without detections
    gt = {"frame": np.asarray(gt["frame"]), "track_id": np.asarray(gt["track_id"]), "ltwh": np.asarray(gt["ltwh"], dtype=np.float64).reshape(-1, 4)}
    eng = HipVideoEngine(pipe)
    df = eng.video_loop(frames, synth_heads=lambda t0, n: heads[t0:t0 + n])
    from_dev = evaluate.evaluate_device_log(gt, eng.last_log, pipe)
    tracked = df[df.track_id.notna()]
    assert len(tracked) > 100
    pred = {"frame": tracked.image_id.to_numpy().astype(np.int64) + 1, "track_id": tracked.track_id.to_numpy().astype(np.int64),
            "ltwh": np.stack(tracked.track_bbox_ltwh.to_list()).astype(np.float64)}
    for device in ("gpu", "cpu"):
        ref = evaluate.evaluate_sequence(gt, pred, n_frames=T, device=device)
        np.testing.assert_allclose(from_dev["hota"], ref["hota"], rtol=1e-12, atol=1e-12, err_msg=device)
        for k, v in ref["clear"].items():
            if isinstance(v, float):
                np.testing.assert_allclose(from_dev["clear"][k], v, rtol=1e-12, atol=1e-12, err_msg=f"{device} {k}")
            else:
                assert from_dev["clear"][k] == v, (device, k, from_dev["clear"][k], v)
    # without fetching the table at all
    assert eng.video_loop(frames, synth_heads=lambda t0, n: heads[t0:t0 + n], fetch_table=False) is None
    again = evaluate.evaluate_device_log(gt, eng.last_log, pipe)
    np.testing.assert_array_equal(again["hota"], from_dev["hota"])
    assert again["clear"] == from_dev["clear"]
    pipe.close()
