"""-m gpu: the fused pipelines at the shapes BASELINE.json's configs name -- YOLOX-m and YOLOX-l detectors, 100 objects, K = 6 part
embeddings with D = 256 (BPBReID) and D = 512 (KPR, kpr.yaml:17,24), the RTMPose-m stage with the OKS motion cost -- and the on-device
modules that round 1 only checked with stand-in backends. For every case: >= 48 frames, detection -> track-id assignment identical to
the oracle chain (C decode/NMS + C tracker fed with the embeddings / keypoints the GPU networks produced)."""
import numpy as np
import pandas as pd
import pytest
from types import SimpleNamespace as NS

pytestmark = pytest.mark.gpu


def _detector_rows(orc, head, ratio):
    boxes, _, _ = orc.yolox_postprocess(head, 640, float(np.float32(ratio)))
    l = np.maximum(0, np.minimum(boxes[:, 0], 1918)).astype(np.float32); t = np.maximum(0, np.minimum(boxes[:, 1], 1078)).astype(np.float32)
    r = np.maximum(1, np.minimum(boxes[:, 2], 1919)).astype(np.float32); b = np.maximum(1, np.minimum(boxes[:, 3], 1079)).astype(np.float32)
    return np.stack([l, t, r - l, b - t], axis=1)


@pytest.mark.parametrize("detector,pose,dim,reid_arch", [("m", None, 256, "resnet50"), ("l", None, 256, "resnet50"), ("m", "m", 512, "resnet50"), ("m", None, 256, "hrnet32")],
                         ids=["config3_yolox_m_bpbreid", "config5_yolox_l_bpbreid", "config4_yolox_m_rtmpose_kpr512_oks", "config3h_hrnet32_reid"])
def test_fused_pipeline_ids_equal_oracle_at_baseline_config_shapes(orc, detector, pose, dim, reid_arch):
    import torch
    from tracklab_amd import gpu_pipeline as gp
    from tracklab_amd.synth import SyntheticStream, render_frame, synth_yolox_head
    F, steps, nobj, maxd = 8, 6, 100, 104
    pipe = gp.DetReidTrackPipeline(detector, n_streams=1, frames_per_step=F, max_dets=maxd, dim=dim, pose=pose, use_graph=False, reid_arch=reid_arch)
    rng = np.random.default_rng(31)
    stream = list(SyntheticStream(41, nobj, F * steps))
    heads = np.stack([synth_yolox_head(rng, fr["dets"][:, :4], ratio=pipe.ratio) for fr in stream])
    pool = np.stack([render_frame(rng, stream[i]["gt_boxes"]) for i in range(F)])              # frame content is irrelevant to the ids: a small pool
    d_frames = torch.from_numpy(pool).cuda()
    d_heads = torch.from_numpy(heads).cuda().reshape(steps, F, -1, 6)
    ref = orc.StrongSORT(pipe.K, pipe.D, **pipe.tracker_cfg)
    assert pipe.tracker_cfg["motion_criterium"] == ("oks" if pose else "iou")
    rows_total, tracks = 0, 0
    for k in range(steps):
        h_rows, h_cnt = pipe.step(d_frames, d_heads[k])
        pipe.synchronize()
        rows, cnt = pipe.rows_numpy(h_rows, h_cnt)
        emb = pipe.last["emb"].cpu().numpy().reshape(F, maxd, pipe.K, pipe.D)
        assert np.isfinite(emb).all()                                      # (fp16 activations of the random-init network stay in range)
        vis = pipe.last["vis"].cpu().numpy().reshape(F, maxd, pipe.K)
        kps = pipe.last["kps"].cpu().numpy().reshape(F, maxd, 17, 3) if pose else None
        for f in range(F):
            ltwh = _detector_rows(orc, heads[k * F + f], pipe.ratio)
            n = len(ltwh)
            assert n == int(pipe.last["counts"][f])
            ids = (k * F + f) * maxd + np.arange(n)
            exp = ref.update(ids, ltwh.astype(np.float64), emb[f, :n], vis[f, :n], np.ones(n), keypoints=None if kps is None else kps[f, :n])
            got = rows[0][f]
            assert len(got) == len(exp), (k, f)
            np.testing.assert_array_equal(got["det_id"], exp["det_id"], err_msg=f"frame {k * F + f}")
            np.testing.assert_array_equal(got["track_id"], exp["track_id"], err_msg=f"frame {k * F + f}")
            np.testing.assert_array_equal(got["matched_name"], exp["matched_name"])
            rows_total += len(exp); tracks = max(tracks, int(exp["track_id"].max()))
    assert rows_total > 0.9 * F * steps * nobj and 100 <= tracks <= 130
    pipe.close()


def test_bpbreid_strongsort_module_on_device_matches_oracle(orc):
    """HipBPBReIDStrongSORT with its real bank (tlk_bpbss through the plugin API), not a stand-in backend."""
    from torch.utils.data.dataloader import default_collate
    from test_modules_host import _frame_df
    from tracklab_amd.synth import SyntheticStream, ltrb_to_ltwh_rows
    from tracklab_amd.wrappers import HipBPBReIDStrongSORT
    K, D = 6, 64
    cfg = NS(ecc=False, ema_alpha=0.9, mc_lambda=0.995, max_dist=0.5, motion_criterium="iou", max_iou_distance=0.8, max_oks_distance=0.7, max_age=300,
             n_init=0, nn_budget=100, min_bbox_confidence=0.0, only_position_for_kf_gating=False, max_kalman_prediction_without_update=7,
             matching_strategy="strong_sort_matching", gating_thres_factor=1, w_kfgd=1, w_reid=1, w_st=1)
    m = HipBPBReIDStrongSORT(cfg, "cuda:0")
    ref = orc.StrongSORT(K, D, **{k: v for k, v in vars(cfg).items() if k != "ecc"})
    n_rows = 0
    for fr in SyntheticStream(14, 40, 50, parts=K, dim=D, with_embeddings=True, miss_prob=0.1, churn_period=10):
        d = fr["dets"]
        df = _frame_df(fr)
        df["embeddings"] = list(fr["embeddings"]); df["visibility_scores"] = list(fr["visibility"])
        out = m.process(default_collate([m.preprocess(None, df, pd.Series({"frame": fr["frame"]}))]), df, None)
        exp = ref.update(d[:, 6].astype(np.int64), ltrb_to_ltwh_rows(d[:, :4]), fr["embeddings"], fr["visibility"], d[:, 4])
        assert list(out.index) == list(exp["det_id"])
        np.testing.assert_array_equal(out.track_id.to_numpy(), exp["track_id"])
        assert list(out.state) == [{0: "t", 1: "c", 2: "d"}[s] for s in exp["state"]]
        np.testing.assert_array_equal(out.hits.to_numpy(), exp["hits"]); np.testing.assert_array_equal(out.age.to_numpy(), exp["age"])
        np.testing.assert_allclose(np.stack(out.track_bbox_kf_ltwh.to_list()), exp["kf_ltwh"], rtol=1e-9, atol=1e-9)
        n_rows += len(exp)
    assert n_rows > 1500


def test_bytetrack_module_on_device_matches_oracle_and_keeps_counting_ids_across_videos(orc):
    from torch.utils.data.dataloader import default_collate
    from test_modules_host import _frame_df
    from tracklab_amd.synth import SyntheticStream
    from tracklab_amd.wrappers import HipByteTrack
    hyper = dict(track_thresh=0.6, track_buffer=30, match_thresh=0.8, frame_rate=30)
    m = HipByteTrack(NS(min_confidence=0.4, hyperparams=hyper), "cuda:0", tracking_dataset=None)
    m.reset()
    offset, max_id = 0, 0
    for video in range(2):                         # the reference's BaseTrack._count is class-level: ids of video 2 continue those of video 1
        ref = orc.ByteTrack(**hyper)
        for fr in SyntheticStream(7 + video, 20, 40, miss_prob=0.1, low_conf_frac=0.3):
            df = _frame_df(fr, np.float64, id0=300)
            out = m.process(default_collate([m.preprocess(None, df, pd.Series({"frame": fr["frame"]}))]), df, None)
            d = fr["dets"].copy(); d[:, 5] = 1.0; d[:, 6] += 300
            exp = ref.update(d[d[:, 4] > 0.4])
            assert len(out) == len(exp)
            if len(exp):
                np.testing.assert_array_equal(out.index.to_numpy(), exp[:, 7].astype(int))
                np.testing.assert_array_equal(out.track_id.to_numpy(), exp[:, 4] + offset)
                np.testing.assert_array_equal(np.stack(out.track_bbox_ltwh.to_list()), np.stack([exp[:, 0], exp[:, 1], exp[:, 2] - exp[:, 0], exp[:, 3] - exp[:, 1]], axis=1))
                max_id = max(max_id, int(out.track_id.max()))
        offset = max_id
        m.reset()
    assert offset > 20


def test_deepocsort_module_on_device_matches_oracle(orc):
    """HipDeepOCSORT: GPU crop + ReID forward + tlk_deepocsort bank through the plugin API; the oracle fed with the module's own features."""
    from torch.utils.data.dataloader import default_collate
    from test_modules_host import _frame_df
    from tracklab_amd.synth import SyntheticStream, render_frame
    from tracklab_amd.wrappers import HipDeepOCSORT
    hyper = dict(det_thresh=0.45, max_age=10, min_hits=1, iou_threshold=0.25, delta_t=2, asso_func="giou", inertia=0.3, w_association_emb=0.75,
                 alpha_fixed_emb=0.95, aw_param=0.5, embedding_off=False, cmc_off=True, aw_off=False, new_kf_off=False)
    D = 64
    m = HipDeepOCSORT(NS(min_confidence=0.4, feature_dim=D, hyperparams=hyper), "cuda:0", tracking_dataset=None)
    ref = orc.DeepOCSort(D, **hyper)
    rng = np.random.default_rng(3)
    feats_seen = []
    orig = m._features
    m._features = lambda image, dets: feats_seen.append(orig(image, dets)) or feats_seen[-1]
    n_rows = 0
    for fr in SyntheticStream(8, 15, 14, miss_prob=0.05, low_conf_frac=0.25):
        img = render_frame(rng, fr["gt_boxes"])
        df = _frame_df(fr, np.float64, id0=200)
        feats_seen.clear()
        sample = m.preprocess(img, df, pd.Series({"frame": fr["frame"]}))
        out = m.process(default_collate([sample]), df, pd.DataFrame({"file_path": ["unused"]}))
        inp = sample["input"]
        keep = inp[:, 4] > 0.4
        use = keep & (inp[:, 4] > 0.45)
        emb = np.zeros((len(inp), D), np.float32)
        if use.any():
            emb[use] = feats_seen[0]
        exp = ref.update(inp[keep], emb[keep])
        assert len(out) == len(exp)
        if len(exp):
            np.testing.assert_array_equal(out.index.to_numpy(), exp[:, 7].astype(int))
            np.testing.assert_array_equal(out.track_id.to_numpy(), exp[:, 4])
            n_rows += len(exp)
    assert n_rows > 60


def test_rtmpose_module_on_device_matches_oracle_pre_and_post_processing(orc):
    """HipRTMPose: the crops the module feeds its network and the keypoints it decodes == oracle rtmlib pre / post-processing around the
    same network outputs (the forward itself is PyTorch-ROCm on both sides of the comparison)."""
    import torch
    from torch.utils.data.dataloader import default_collate
    from tracklab_amd.synth import SyntheticStream, render_frame
    from tracklab_amd.wrappers import HipRTMPose
    m = HipRTMPose("cuda:0", cfg=NS(arch="m", model_input_size=[192, 256], max_dets=32), tracking_dataset=None)
    rng = np.random.default_rng(4)
    fr = SyntheticStream(9, 12, 1).step()
    img = render_frame(rng, fr["gt_boxes"])
    d = fr["dets"]
    ltwh = np.column_stack([d[:, 0], d[:, 1], d[:, 2] - d[:, 0], d[:, 3] - d[:, 1]]).astype(np.float32)
    ltwh[0] = [0, 5, 60, 150]; ltwh[1] = [1850, 900, 69, 179]           # boxes touching the frame border (sanitized like the detector's)
    df = pd.DataFrame({"bbox_ltwh": list(ltwh)}, index=np.arange(50, 50 + len(ltwh)))
    captured = {}
    m._ensure_model()
    net = m._model
    m._model = lambda crops: captured.setdefault("out", (captured.setdefault("crops", crops.clone()), net(crops))[1])
    out = m.process(default_collate([m.preprocess(img, df, pd.Series(dtype=float))]), df, pd.DataFrame())
    sx, sy = captured["out"]
    crops = captured["crops"].float().cpu().numpy()                     # (n, 3, 256, 192) logical NCHW
    bgr = np.ascontiguousarray(img[..., ::-1])                          # the reference's cv2.imread frame; the module reads the RGB frame as BGR
    for i in range(len(df)):
        l, t, w, h = (float(v) for v in ltwh[i])
        xyxy = np.array([l, t, np.float32(l) + np.float32(w), np.float32(t) + np.float32(h)], dtype=np.float64)
        crop, c, sc = orc.rtmpose_preprocess(bgr, xyxy)
        np.testing.assert_array_equal(crops[i], torch.from_numpy(crop).half().float().numpy())
        kp, score = orc.simcc_decode(sx[i].float().cpu().numpy(), sy[i].float().cpu().numpy(), c, sc)
        got = out.keypoints_xyc.iloc[i]
        np.testing.assert_allclose(got[:, :2], kp, rtol=1e-12, atol=1e-9)
        np.testing.assert_array_equal(got[:, 2].astype(np.float32), score.astype(np.float32))


def test_detector_to_tracker_path_with_the_networks_own_head_activations(orc):
    """the bench replaces the random-init detector's head activations by a synthetic head (disclosed in its line), and nothing tested
    frames -> letterbox -> forward -> decode + NMS -> tracker with the activations the network itself produced. Here the head's objectness / class
    biases are raised so that a random-init YOLOX-s fires on a few hundred anchors per frame, no synthetic head is passed, and every stage after the
    forward is replayed by the oracle from the SAME head tensor: detections (count, class, score exact; boxes to the 1-2 ulp of expf) and the
    OC-SORT rows the fused step hands out (ids exact)."""
    import torch
    from tracklab_amd import gpu_pipeline as gp
    from tracklab_amd.synth import SyntheticStream, render_frame
    torch.manual_seed(3)
    F = 4
    pipe = gp.DetTrackPipeline("s", n_streams=1, frames_per_step=F, max_dets=128, use_graph=False)
    head = pipe.model.head
    rng = np.random.default_rng(5)
    stream = SyntheticStream(9, 30, 3 * F + F)
    from tracklab_amd import _lib
    calib = torch.from_numpy(np.stack([render_frame(rng, stream.step()["gt_boxes"]) for _ in range(F)])).cuda()
    x, _ = _lib.letterbox(calib, pipe.size, pipe.layout, pipe.dtype, swap_rb=True)
    chosen = None
    with torch.no_grad():
        for k in range(len(head.obj_preds)):
            head.reg_preds[k].bias.copy_(torch.tensor([0.0, 0.0, 1.2, 1.6]))         # boxes of ~3.3 x 5 cells: neighbours overlap, NMS has work to do
        for bias in np.arange(0.6, 4.01, 0.1):                                        # the lowest bias at which every frame has 60 .. 3000 candidates
            for k in range(len(head.obj_preds)):
                head.obj_preds[k].bias.fill_(float(bias)); head.cls_preds[k].bias.fill_(float(bias))
            p = pipe.model(x, focused=(pipe.layout == "focus_nhwc")).float()
            cand = ((p[..., 4] * p[..., 5]) > 0.7).sum(dim=1)
            if int(cand.min()) >= 60:
                chosen = (float(bias), cand.cpu().tolist())
                break
    assert chosen is not None and max(chosen[1]) <= 3000, chosen
    stash = {}
    fwd = pipe.model.forward

    def spy(x, **kw):
        stash["pred"] = fwd(x, **kw)
        return stash["pred"]
    pipe.model.forward = spy
    ref = orc.OCSort(**pipe.tracker_cfg["hyper"])
    total = 0
    for step in range(3):
        frames = np.stack([render_frame(rng, stream.step()["gt_boxes"]) for _ in range(F)])
        pipe.step(torch.from_numpy(frames).cuda())
        pipe.synchronize()
        pred = stash["pred"].float().cpu().numpy()
        assert np.isfinite(pred).all()
        res = pipe.host_results(pipe.last)
        det = pipe.last["det"]
        counts = det["counts"].cpu().numpy()
        for f in range(F):
            eb, es, ec = orc.yolox_postprocess(pred[f], 640, float(np.float32(pipe.ratio)))
            eb, es, ec = eb[:pipe.maxd], es[:pipe.maxd], ec[:pipe.maxd]
            n = int(counts[f])
            assert n == len(eb) and n > 0, (step, f, n, len(eb))
            total += n
            np.testing.assert_array_equal(det["cls"][f, :n].cpu().numpy(), ec)
            np.testing.assert_array_equal(det["scores"][f, :n].cpu().numpy(), es)
            np.testing.assert_allclose(det["xyxy"][f, :n].cpu().numpy(), eb, rtol=2e-6, atol=1e-4)
            # the tracker is fed with the GPU's own detector rows (their last bits are expf's): its rows must be the oracle's on those rows
            trk_in = pipe.last["trk_in"].view(F, pipe.maxd, 7)[f, :n].cpu().numpy()
            exp = orc.ocsort_wrapper_step(ref, trk_in, pipe.tracker_cfg["min_confidence"])
            got = res["rows"][f, :int(res["ocnt"][f])]
            assert got.shape == exp.shape, (step, f, got.shape, exp.shape)
            np.testing.assert_array_equal(got[:, [4, 5, 7]], exp[:, [4, 5, 7]])
            np.testing.assert_array_equal(got, exp)
    assert total > 40
    pipe.close()


def _ocsort_dets(ltwh, ids):
    d = np.zeros((len(ltwh), 7))
    d[:, 0], d[:, 1] = ltwh[:, 0], ltwh[:, 1]
    d[:, 2], d[:, 3] = (ltwh[:, 0] + ltwh[:, 2]).astype(np.float32), (ltwh[:, 1] + ltwh[:, 3]).astype(np.float32)
    d[:, 4], d[:, 5], d[:, 6] = 1.0, 1.0, ids
    return d


@pytest.mark.parametrize("use_graph", [False, True], ids=["eager", "graph"])
def test_config2_fused_pipeline_at_its_baseline_size(orc, use_graph):
    """BASELINE.json configs[1] EXACTLY -- YOLOX-s + OC-SORT on a synthetic 1080p 50-object stream, 32 frames per
    step, max_dets 128 (bench.py WORKLOADS["config2"]) -- through the fused DetTrackPipeline: detection -> track id identical to the oracle
    chain (C decode / NMS + C OC-SORT) on every frame of 3 steps."""
    import torch
    from tracklab_amd import gpu_pipeline as gp
    from tracklab_amd.synth import SyntheticStream, render_frame, synth_yolox_head
    F, steps, nobj, maxd = 32, 3, 50, 128
    pipe = gp.DetTrackPipeline("s", n_streams=1, frames_per_step=F, max_dets=maxd, use_graph=use_graph)
    rng = np.random.default_rng(77)
    stream = list(SyntheticStream(2, nobj, F * steps))
    heads = np.stack([synth_yolox_head(rng, fr["dets"][:, :4], ratio=pipe.ratio) for fr in stream])
    d_frames = torch.from_numpy(np.stack([render_frame(rng, stream[i]["gt_boxes"]) for i in range(F)])).cuda()
    d_heads = torch.from_numpy(heads).cuda().reshape(steps, F, -1, 6)
    ref = orc.OCSort(**pipe.tracker_cfg["hyper"])
    rows_total = 0
    for k in range(steps):
        rows, cnt = pipe.step(d_frames, d_heads[k])
        pipe.synchronize()
        rows_a = pipe.rows_array(rows)
        for f in range(F):
            ltwh = _detector_rows(orc, heads[k * F + f], pipe.ratio)
            exp = orc.ocsort_wrapper_step(ref, _ocsort_dets(ltwh, (k * F + f) * maxd + np.arange(len(ltwh))), pipe.tracker_cfg["min_confidence"])
            got = np.array(rows_a[0, f, :int(cnt[0, f])])
            assert got.shape == exp.shape, (k, f, got.shape, exp.shape)
            np.testing.assert_array_equal(got[:, [4, 7]], exp[:, [4, 7]], err_msg=f"frame {k * F + f}")
            rows_total += len(exp)
    assert rows_total > 0.85 * F * steps * nobj
    pipe.close()


def test_config5_unit_eight_streams_resident_on_one_gpu(orc):
    """the per-GPU unit of BASELINE.json configs[4] (YOLOX-l + part-based ReID + BPBReID-StrongSORT) with EIGHT tracker banks
    resident on one GPU -- n_streams = 8, one frame of every stream per step (the online shape of the 8-stream job folded onto one device):
    every stream's detection -> track id assignment equals its own oracle chain, i.e. the streams do not leak into each other."""
    import torch
    from tracklab_amd import gpu_pipeline as gp
    from tracklab_amd.synth import SyntheticStream, render_frame, synth_yolox_head
    S, F, steps, nobj, maxd = 8, 1, 10, 40, 48
    pipe = gp.DetReidTrackPipeline("l", n_streams=S, frames_per_step=F, max_dets=maxd, use_graph=True)
    rng = np.random.default_rng(8)
    streams = [list(SyntheticStream(100 + s, nobj, steps)) for s in range(S)]
    heads = np.stack([[synth_yolox_head(rng, streams[s][k]["dets"][:, :4], ratio=pipe.ratio) for s in range(S)] for k in range(steps)])     # (steps, S, A, 6)
    d_frames = torch.from_numpy(np.stack([render_frame(rng, streams[s][0]["gt_boxes"]) for s in range(S)])).cuda()
    d_heads = torch.from_numpy(heads).cuda()
    refs = [orc.StrongSORT(pipe.K, pipe.D, **pipe.tracker_cfg) for _ in range(S)]
    tracks = [0] * S
    for k in range(steps):
        h_rows, h_cnt = pipe.step(d_frames, d_heads[k])
        pipe.synchronize()
        rows, cnt = pipe.rows_numpy(h_rows, h_cnt)
        emb = pipe.last["emb"].cpu().numpy().reshape(S, F, maxd, pipe.K, pipe.D)
        vis = pipe.last["vis"].cpu().numpy().reshape(S, F, maxd, pipe.K)
        for s in range(S):
            ltwh = _detector_rows(orc, heads[k, s], pipe.ratio)
            n = len(ltwh)
            ids = (k * S * F + s * F) * maxd + np.arange(n)
            exp = refs[s].update(ids, ltwh.astype(np.float64), emb[s, 0, :n], vis[s, 0, :n], np.ones(n))
            got = rows[s][0]
            assert len(got) == len(exp), (k, s)
            np.testing.assert_array_equal(got["det_id"], exp["det_id"], err_msg=f"step {k} stream {s}")
            np.testing.assert_array_equal(got["track_id"], exp["track_id"], err_msg=f"step {k} stream {s}")
            tracks[s] = max(tracks[s], int(exp["track_id"].max()) if len(exp) else 0)
    assert all(nobj - 4 <= t <= nobj + 12 for t in tracks), tracks          # every bank numbers ITS OWN tracks from 1
    pipe.close()
