#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ by IMPORTING the reference.

Runs only in the build container (needs /root/reference). Nothing here travels to
the GPU box except the small ``.npz`` / ``.json`` fixtures it writes. Recipe =
SURVEY.md Appendix A:

* ``plugins/track/oc_sort`` imports with a 3-name ``filterpy`` shim
  (``oc_sort/kalmanfilter.py:105-106``); ``lap`` is absent so the reference's own
  scipy fallback runs (``oc_sort/association.py:187-195``).
* ``plugins/track/bpbreid_strong_sort`` imports with a stub ``cv2`` (``ecc.py:4``)
  and with ``compute_distance_matrix_using_bp_features`` injected into
  ``sort/nn_matching.py`` (its try/except at ``:4-8`` leaves the name undefined).
  That function is third-party (torchreid fork, unpinned): restated below from its
  published algorithm -> parity is UNPINNED at that single call.

Usage:  python tests/golden/make_golden.py   (solver recorded: scipy <version>)
"""
from __future__ import annotations

import hashlib
import json
import os
import sys
import types

import numpy as np
import pandas as pd
import scipy
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REF, "plugins", "track"))
sys.path.insert(0, REF)

from tracklab_amd.synth import SyntheticStream, ltrb_to_ltwh_rows, synth_keypoints  # noqa: E402


# ----------------------------------------------------------------------------- shims
def _install_filterpy_shim():
    fp = types.ModuleType("filterpy")
    st = types.ModuleType("filterpy.stats")
    cm = types.ModuleType("filterpy.common")
    st.logpdf = lambda *a, **k: 0.0
    cm.pretty_str = lambda label, arr: f"{label} = {arr}"

    def reshape_z(z, dim_z, ndim):
        z = np.atleast_2d(z)
        if z.shape[1] == dim_z:
            z = z.T
        if z.shape != (dim_z, 1):
            raise ValueError("z must be convertible to shape ({}, 1)".format(dim_z))
        if ndim == 1:
            z = z[:, 0]
        if ndim == 0:
            z = z[0, 0]
        return z

    cm.reshape_z = reshape_z
    fp.stats, fp.common = st, cm
    sys.modules.update({"filterpy": fp, "filterpy.stats": st, "filterpy.common": cm})


def _install_cv2_stub():
    cv2 = types.ModuleType("cv2")
    cv2.MOTION_EUCLIDEAN = 1
    cv2.MOTION_HOMOGRAPHY = 3
    sys.modules["cv2"] = cv2


def bp_distance_restated(qf, gf, qvis, gvis, use_gpu=False):
    """Restatement of torchreid(fork).metrics.distance.compute_distance_matrix_using_bp_features
    (boolean-visibility branch, 'mean' combine, euclidean metric): per part
    sqrt(relu(|q|^2 - 2 q.g + |g|^2)), mean over parts visible in both, -1 if none."""
    q = qf.transpose(1, 0)
    g = gf.transpose(1, 0)
    dot = torch.matmul(q, g.transpose(2, 1))
    qs = q.pow(2).sum(dim=-1)
    gs = g.pow(2).sum(dim=-1)
    d = qs.unsqueeze(2) - 2 * dot + gs.unsqueeze(1)
    d = torch.sqrt(torch.nn.functional.relu(d))
    valid = qvis.t().unsqueeze(2) * gvis.t().unsqueeze(1)
    validf = valid.to(d.dtype)
    cnt = validf.sum(dim=0)
    pair = (d * validf).sum(dim=0) / cnt.clamp(min=1)
    pair = torch.where(cnt == 0, torch.full_like(pair, -1.0), pair)
    return pair, d


def sha(*arrays) -> str:
    h = hashlib.sha256()
    for a in arrays:
        h.update(np.ascontiguousarray(a).tobytes())
    return h.hexdigest()


# ----------------------------------------------------------------------------- OC-SORT
OCSORT_CONFIGS = {
    # tracklab/configs/modules/track/oc_sort.yaml:3-14
    "yaml": dict(min_confidence=0.4, hyper=dict(asso_func="giou", delta_t=1, det_thresh=0,
                 inertia=0.3941737016672115, iou_threshold=0.22136877277096445,
                 max_age=50, min_hits=1, use_byte=False)),
    # OCSort.__init__ defaults (ocsort.py:186-187)
    "defaults": dict(min_confidence=0.4, hyper=dict(asso_func="iou", delta_t=3, det_thresh=0.6,
                     inertia=0.2, iou_threshold=0.3, max_age=30, min_hits=3, use_byte=False)),
    # BASELINE.json configs[0]: "IoU-only SORT" = inertia 0, asso_func iou
    "iou_only": dict(min_confidence=0.4, hyper=dict(asso_func="iou", delta_t=3, det_thresh=0,
                     inertia=0.0, iou_threshold=0.3, max_age=30, min_hits=3, use_byte=False)),
    "byte_diou": dict(min_confidence=0.0, hyper=dict(asso_func="diou", delta_t=2, det_thresh=0.5,
                      inertia=0.2, iou_threshold=0.3, max_age=20, min_hits=2, use_byte=True)),
    "ciou": dict(min_confidence=0.4, hyper=dict(asso_func="ciou", delta_t=3, det_thresh=0.1,
                 inertia=0.3, iou_threshold=0.25, max_age=10, min_hits=1, use_byte=False)),
    "ct": dict(min_confidence=0.4, hyper=dict(asso_func="ct_dist", delta_t=3, det_thresh=0.1,
               inertia=0.1, iou_threshold=0.3, max_age=15, min_hits=1, use_byte=False)),
}

OCSORT_RUNS = [  # (name, config, seed, n_objects, n_frames, stream kwargs)
    ("yaml_s0_n100", "yaml", 0, 100, 200, {}),
    ("yaml_s1_n50", "yaml", 1, 50, 200, {}),
    ("yaml_s2_n10", "yaml", 2, 10, 200, {"miss_prob": 0.15, "churn_period": 20}),
    ("yaml_s3_n30_gaps", "yaml", 3, 30, 200, {"miss_prob": 0.3, "churn_period": 10}),
    ("defaults_s0_n50", "defaults", 0, 50, 200, {"miss_prob": 0.1}),
    ("iou_only_s1_n30", "iou_only", 1, 30, 150, {"miss_prob": 0.1}),
    ("byte_s2_n40", "byte_diou", 2, 40, 150, {"miss_prob": 0.1, "low_conf_frac": 0.25}),
    ("ciou_s4_n20", "ciou", 4, 20, 120, {"miss_prob": 0.2, "churn_period": 10}),
    ("ct_s5_n20", "ct", 5, 20, 120, {"miss_prob": 0.1, "churn_period": 10}),
    ("yaml_s6_n100_cls2", "yaml", 6, 100, 100, {"cls": 2.0, "miss_prob": 0.05}),
    ("yaml_s7_n120_crowded", "yaml", 7, 120, 120, {"miss_prob": 0.1, "churn_period": 15}),                     # r03: the crowded regime, LSA path every frame
]


def gen_ocsort(out_dir):
    _install_filterpy_shim()
    import oc_sort.ocsort as ref  # noqa

    for name, cfgname, seed, nobj, nframes, skw in OCSORT_RUNS:
        cfg = OCSORT_CONFIGS[cfgname]
        tracker = ref.OCSort(**cfg["hyper"])
        stream = SyntheticStream(seed, nobj, nframes, **skw)
        captured = {}

        real_associate = ref.associate

        def spy(dets, trks, thr, vel, kobs, w, _real=real_associate, _cap=captured):
            r = _real(dets, trks, thr, vel, kobs, w)
            _cap["matched"] = np.asarray(r[0], dtype=np.int64).reshape(-1, 2)
            _cap["um_dets"] = np.asarray(r[1], dtype=np.int64)
            _cap["um_trks"] = np.asarray(r[2], dtype=np.int64).reshape(-1)
            if len(trks):
                _cap["iou"] = ref.iou_batch(dets, trks)
            return r

        ref.associate = spy
        blobs = {}
        in_off, out_off = [0], [0]
        ins, outs = [], []
        for fr in stream:
            dets = fr["dets"]
            if fr["frame"] % 37 == 5:        # a few empty frames: wrapper skips the tracker (oc_sort_api.py:51-52)
                dets = dets[:0]
            ins.append(dets)
            in_off.append(in_off[-1] + len(dets))
            if len(dets) == 0:
                res = np.empty((0, 8))
            else:
                inp = torch.from_numpy(dets)
                inp = inp[inp[:, 4] > cfg["min_confidence"]]
                captured.clear()
                res = np.asarray(tracker.update(inp, None), dtype=np.float64).reshape(-1, 8) \
                    if True else None
                f = fr["frame"]
                if f in (1, 7, 50, 99, 120) and "matched" in captured:
                    blobs[f"f{f}_matched"] = captured["matched"]
                    blobs[f"f{f}_um_dets"] = captured["um_dets"]
                    blobs[f"f{f}_um_trks"] = captured["um_trks"]
                    if "iou" in captured:
                        blobs[f"f{f}_iou"] = captured["iou"]
                if f in (7, 50, 99, 119, 149, 199):
                    blobs[f"f{f}_kf_x"] = np.stack([t.kf.x[:, 0] for t in tracker.trackers]) \
                        if tracker.trackers else np.empty((0, 7))
                    blobs[f"f{f}_kf_P"] = np.stack([t.kf.P for t in tracker.trackers]) \
                        if tracker.trackers else np.empty((0, 7, 7))
                    blobs[f"f{f}_ids"] = np.array([t.id for t in tracker.trackers], dtype=np.int64)
            outs.append(res)
            out_off.append(out_off[-1] + len(res))
        ref.associate = real_associate
        np.savez_compressed(
            os.path.join(out_dir, f"ocsort_{name}.npz"),
            dets=np.concatenate(ins) if ins else np.empty((0, 7)),
            det_offsets=np.array(in_off, dtype=np.int64),
            out=np.concatenate(outs), out_offsets=np.array(out_off, dtype=np.int64),
            config=json.dumps(cfg), seed=seed, n_objects=nobj, n_frames=nframes,
            stream_kwargs=json.dumps(skw), **blobs)
        print(f"ocsort_{name}: frames={nframes} rows_out={out_off[-1]} ids={int(ref.KalmanBoxTracker.count)}")


def gen_iou_family(out_dir):
    _install_filterpy_shim()
    import oc_sort.association as A
    rng = np.random.default_rng(11)
    blobs = {}
    for tag, (n, m) in {"a": (17, 23), "b": (64, 64), "c": (1, 5), "d": (100, 130)}.items():
        def boxes(k):
            x = rng.uniform(0, 1800, k)
            y = rng.uniform(0, 1000, k)
            w = rng.uniform(5, 200, k)
            h = rng.uniform(5, 300, k)
            return np.stack([x, y, x + w, y + h], 1)
        b1, b2 = boxes(n), boxes(m)
        if n > 3:
            b2[:3] = b1[:3] + rng.normal(0, 2, (3, 4))        # some heavy overlaps
        blobs[f"{tag}_b1"], blobs[f"{tag}_b2"] = b1, b2
        for fn in ("iou_batch", "giou_batch", "diou_batch", "ciou_batch", "ct_dist"):
            blobs[f"{tag}_{fn}"] = getattr(A, fn)(b1, b2)
    np.savez_compressed(os.path.join(out_dir, "iou_family.npz"), **blobs)
    print("iou_family ok")


def gen_kf7(out_dir):
    """KalmanBoxTracker / KalmanFilterNew unit vectors incl. freeze/unfreeze replays
    (ocsort.py:57-169, kalmanfilter.py:339-526)."""
    _install_filterpy_shim()
    import oc_sort.ocsort as ref
    rng = np.random.default_rng(5)
    blobs = {}
    case = 0
    for gap_pattern in ([1] * 12, [1, 1, 0, 1, 1], [1, 0, 0, 1, 0, 1, 1], [1, 1, 0, 0, 0, 0, 0, 1, 1, 0, 1],
                        [0, 0, 1, 1, 0, 1], [1, 0, 1, 0, 1, 0, 0, 0, 1]):
        for rep in range(3):
            ref.KalmanBoxTracker.count = 0
            x0, y0 = rng.uniform(200, 1500), rng.uniform(200, 800)
            w, h = rng.uniform(40, 120), rng.uniform(80, 300)
            vx, vy = rng.normal(0, 4), rng.normal(0, 3)
            box = lambda t: np.array([x0 + vx * t - w / 2, y0 + vy * t - h / 2,
                                      x0 + vx * t + w / 2, y0 + vy * t + h / 2, 0.9]) \
                + np.r_[rng.normal(0, 1, 4), 0]
            b0 = box(0)
            trk = ref.KalmanBoxTracker(b0, 1.0, delta_t=3, tracklab_id=0.0)
            obs, xs, Ps, preds, vels = [b0], [], [], [], []
            for t, seen in enumerate(gap_pattern, start=1):
                preds.append(trk.predict()[0])
                if seen:
                    b = box(t)
                    trk.update(b, 1.0, float(t))
                    obs.append(b)
                else:
                    trk.update(None, None)
                    obs.append(np.full(5, np.nan))
                xs.append(trk.kf.x[:, 0].copy())
                Ps.append(trk.kf.P.copy())
                vels.append(np.zeros(2) if trk.velocity is None else np.asarray(trk.velocity, dtype=float))
            blobs[f"c{case}_pattern"] = np.array(gap_pattern, dtype=np.int64)
            blobs[f"c{case}_obs"] = np.stack(obs)
            blobs[f"c{case}_x"] = np.stack(xs)
            blobs[f"c{case}_P"] = np.stack(Ps)
            blobs[f"c{case}_pred"] = np.stack(preds)
            blobs[f"c{case}_vel"] = np.stack(vels)
            case += 1
    blobs["n_cases"] = np.int64(case)
    np.savez_compressed(os.path.join(out_dir, "kf7_cases.npz"), **blobs)
    print("kf7 cases", case)


# ----------------------------------------------------------------------------- LSA
def gen_lsa(out_dir):
    from scipy.optimize import linear_sum_assignment
    rng = np.random.default_rng(3)
    cases = []

    def add(c):
        c = np.asarray(c, dtype=np.float64)
        r, cc = linear_sum_assignment(c)
        cases.append((c, r.astype(np.int64), cc.astype(np.int64)))

    # known-answer matrices held by the reference's vendored py-motmetrics tests
    # (plugins/eval/PoseTrack21/posetrack21_mot/posetrack21_mot/motmetrics/tests/test_lap.py:31-176),
    # finite ones only; expected values are asserted in tests/test_lsa.py
    kat = [
        ([[6, 9, 1], [10, 3, 2], [8, 7, 4]], [0, 1, 2], [2, 1, 0]),
        ([[5, 5, 6], [1, 2, 5], [2, 4, 5]], [0, 1, 2], [2, 1, 0]),
        ([[-2, -2, -1], [-6, -5, -2], [-5, -3, -2]], [0, 1, 2], [2, 1, 0]),
        ([[6, 4, 1], [10, 8, 2]], [0, 1], [1, 2]),
        ([[6, 10], [4, 8], [1, 2]], [1, 2], [0, 1]),
    ]
    with open(os.path.join(out_dir, "lsa_kat.json"), "w") as f:
        json.dump({"source": "motmetrics/tests/test_lap.py:31-176 (finite cases)",
                   "cases": [{"cost": c, "rows": r, "cols": cc} for c, r, cc in kat]}, f, indent=1)
    for (n, m) in [(1, 1), (1, 7), (7, 1), (5, 5), (8, 13), (13, 8), (40, 40), (64, 65), (65, 64),
                   (100, 100), (100, 117), (117, 100), (128, 128), (3, 200), (200, 3)]:
        add(rng.uniform(0, 1, (n, m)))
        add(-rng.uniform(0, 1, (n, m)))
        # tie-heavy: clamped like linear_assignment.py:55 and integer like GT boxes
        c = rng.uniform(0, 1, (n, m))
        c[c > 0.5] = 0.5 + 1e-5
        add(c)
        add(rng.integers(0, 4, (n, m)).astype(float))
        add(np.zeros((n, m)))
        c = rng.uniform(0, 1, (n, m))
        c[rng.uniform(0, 1, (n, m)) < 0.7] = 1e5
        add(c)
    blobs = {"n_cases": np.int64(len(cases)), "scipy_version": scipy.__version__}
    for i, (c, r, cc) in enumerate(cases):
        blobs[f"c{i}_cost"], blobs[f"c{i}_rows"], blobs[f"c{i}_cols"] = c, r, cc
    np.savez_compressed(os.path.join(out_dir, "lsa_cases.npz"), **blobs)
    print("lsa cases", len(cases))


# ----------------------------------------------------------------------------- coordinates
def gen_coords(out_dir):
    import tracklab.utils.coordinates as C
    rng = np.random.default_rng(9)
    boxes = np.concatenate([
        rng.uniform(-50, 2000, (40, 4)),
        np.array([[0, 0, 10, 10], [1915, 1075, 30, 30], [-5, -5, 3, 3], [100.5, 200.5, 50.49, 60.51],
                  [1919, 1079, 1, 1], [10, 10, 0, 0]], dtype=float)])
    blobs = {"boxes": boxes}
    shape = (1920, 1080)
    for fn in ("ltwh_to_ltrb", "ltrb_to_ltwh", "ltwh_to_xywh", "ltrb_to_xywh", "xywh_to_ltrb", "xywh_to_ltwh"):
        f = getattr(C, fn)
        blobs[fn] = np.stack([f(b.copy()) for b in boxes])
        if fn.startswith("xywh_to"):   # the reference recurses forever here (coordinates.py:345<->374)
            continue
        blobs[fn + "_clip"] = np.stack([f(b.copy(), shape) for b in boxes])
        blobs[fn + "_clip_round"] = np.stack([f(b.copy(), shape, True) for b in boxes]).astype(np.int64)
    np.savez_compressed(os.path.join(out_dir, "coords.npz"), **blobs)
    print("coords ok")


# ----------------------------------------------------------------------------- BPBReID-StrongSORT
BPB_YAML = dict(  # tracklab/configs/modules/track/bpbreid_strong_sort.yaml:3-21
    ema_alpha=0.9, mc_lambda=0.995, max_dist=0.5, motion_criterium="iou", max_iou_distance=0.8,
    max_oks_distance=0.7, max_age=300, n_init=0, nn_budget=100, min_bbox_confidence=0.0,
    only_position_for_kf_gating=False, max_kalman_prediction_without_update=7,
    matching_strategy="strong_sort_matching", gating_thres_factor=1, w_kfgd=1, w_reid=1, w_st=1)

BPB_RUNS = [  # (name, cfg overrides, seed, n_objects, n_frames, K, D, store_inputs, stream kwargs)
    ("yaml_s0_n100_d256", {}, 0, 100, 40, 6, 256, False, {}),
    ("yaml_s1_n50_d256", {}, 1, 50, 80, 6, 256, False, {"miss_prob": 0.05}),
    ("yaml_s2_n12_d16", {}, 2, 12, 200, 6, 16, True, {"miss_prob": 0.2, "churn_period": 15}),
    ("ninit3_s3_n20_d32", {"n_init": 3, "max_age": 20, "min_bbox_confidence": 0.55, "max_dist": 0.35},
     3, 20, 150, 6, 32, True, {"miss_prob": 0.2, "churn_period": 10}),
    ("posonly_s4_n20_d32", {"only_position_for_kf_gating": True, "max_kalman_prediction_without_update": 2,
                            "max_age": 30, "ema_alpha": 0.5},
     4, 20, 120, 4, 32, True, {"miss_prob": 0.25, "churn_period": 10}),
    ("botsort_s5_n20_d32", {"matching_strategy": "bot_sort_matching", "max_age": 30, "n_init": 1,
                            "gating_thres_factor": 1.5},
     5, 20, 120, 6, 32, True, {"miss_prob": 0.15, "churn_period": 10}),
    ("yaml_s6_n100_d512", {}, 6, 100, 20, 6, 512, False, {}),
    # motion_criterium "oks": stage B (and BoT-SORT's spatio-temporal term) on COCO keypoints (sort/oks_matching.py)
    ("oks_s7_n20_d32", {"motion_criterium": "oks", "max_oks_distance": 0.7, "n_init": 2, "max_age": 30, "max_dist": 0.3},
     7, 20, 120, 6, 32, True, {"miss_prob": 0.2, "churn_period": 10}),
    ("oks_botsort_s8_n15_d32", {"motion_criterium": "oks", "matching_strategy": "bot_sort_matching", "max_age": 30, "n_init": 1},
     8, 15, 100, 6, 32, True, {"miss_prob": 0.15, "churn_period": 10}),
    # crowded scene with the yaml's max_age 300: track indices in the hundreds, i.e. far above the 8- / 32-entry tables of the small
    # sets behind `list(set(track_indices) - matched)` (sort/linear_assignment.py:128) -- CPython's set order decides the IoU-stage rows
    ("crowded_s100_n135_d32", {}, 100, 135, 60, 6, 32, False, {"miss_prob": 0.15, "churn_period": 3}),
]

STATE_CODE = {"t": 0, "c": 1, "d": 2}


def gen_bpbss(out_dir):
    _install_cv2_stub()
    import bpbreid_strong_sort.sort.nn_matching as nnm
    nnm.compute_distance_matrix_using_bp_features = bp_distance_restated
    import bpbreid_strong_sort.strong_sort as ss
    import bpbreid_strong_sort.sort.tracker as trk_mod  # noqa

    for name, over, seed, nobj, nframes, K, D, store, skw in BPB_RUNS:
        if os.environ.get("GOLDEN_ONLY_RUN") and os.environ["GOLDEN_ONLY_RUN"] != name:
            continue
        cfg = dict(BPB_YAML)
        cfg.update(over)
        model = ss.StrongSORT(**cfg)
        stream = SyntheticStream(seed, nobj, nframes, parts=K, dim=D, with_embeddings=True, **skw)
        in_off, out_off = [0], [0]
        ltwhs, confs, ids_in, embs, viss = [], [], [], [], []
        o_idx, o_tid, o_kf, o_pred, o_pred_valid, o_mname, o_mdist = [], [], [], [], [], [], []
        o_hits, o_age, o_tsu, o_state = [], [], [], []
        h = hashlib.sha256()
        blobs = {}
        use_kp = cfg.get("motion_criterium") == "oks"
        kp_rng = np.random.default_rng(900 + seed)
        kps = []
        for fr in stream:
            dets = fr["dets"]
            if fr["frame"] % 41 == 7:
                dets, emb, vis = dets[:0], fr["embeddings"][:0], fr["visibility"][:0]
            else:
                emb, vis = fr["embeddings"], fr["visibility"]
            kp = synth_keypoints(kp_rng, dets[:, :4]) if use_kp else None
            if use_kp:
                kps.append(kp)
            ltwh = ltrb_to_ltwh_rows(dets[:, :4])
            conf = dets[:, 4].copy()
            did = dets[:, 6].astype(np.int64)
            h.update(np.ascontiguousarray(ltwh).tobytes())
            h.update(np.ascontiguousarray(conf).tobytes())
            h.update(np.ascontiguousarray(emb).tobytes())
            h.update(np.ascontiguousarray(vis).tobytes())
            ltwhs.append(ltwh), confs.append(conf), ids_in.append(did)
            if store:
                embs.append(emb), viss.append(vis)
            in_off.append(in_off[-1] + len(dets))
            n_out = 0
            if len(dets) > 0:       # wrapper: process() returns [] on an empty frame (bpbreid_strong_sort_api.py:103-104)
                df = model.update(torch.from_numpy(did), torch.from_numpy(ltwh), torch.from_numpy(emb),
                                  torch.from_numpy(vis), torch.from_numpy(conf),
                                  torch.zeros(len(dets), dtype=torch.float64),
                                  torch.ones(len(dets), dtype=torch.float64) * fr["frame"],
                                  torch.from_numpy(kp) if use_kp else None)
                for det_id, row in df.iterrows():
                    o_idx.append(int(det_id))
                    o_tid.append(int(row.track_id))
                    o_kf.append(np.asarray(row.track_bbox_kf_ltwh, dtype=np.float64))
                    if row.track_bbox_pred_kf_ltwh is None:
                        o_pred.append(np.full(4, np.nan)), o_pred_valid.append(0)
                    else:
                        o_pred.append(np.asarray(row.track_bbox_pred_kf_ltwh, dtype=np.float64))
                        o_pred_valid.append(1)
                    if row.matched_with is None:
                        o_mname.append(0), o_mdist.append(np.nan)
                    else:
                        o_mname.append({"R": 1, "S": 2}[row.matched_with[0]])
                        o_mdist.append(float(row.matched_with[1]))
                    o_hits.append(int(row.hits)), o_age.append(int(row.age))
                    o_tsu.append(int(row.time_since_update)), o_state.append(STATE_CODE[row.state])
                    n_out += 1
                f = fr["frame"]
                if f in (3, 10, 25, 39, 60, 119) and len(model.tracker.tracks):
                    tr = model.tracker.tracks
                    blobs[f"f{f}_track_ids"] = np.array([t.track_id for t in tr], dtype=np.int64)
                    blobs[f"f{f}_mean"] = np.stack([t.mean for t in tr])
                    blobs[f"f{f}_cov"] = np.stack([t.covariance for t in tr])
                    blobs[f"f{f}_feat"] = np.stack([t.features[-1]["reid_features"] for t in tr]).astype(np.float32)
                    blobs[f"f{f}_fvis"] = np.stack([np.asarray(t.features[-1]["visibility_scores"]) for t in tr])
            out_off.append(out_off[-1] + n_out)
        extra = {}
        if store:
            extra["embeddings"] = np.concatenate(embs).astype(np.float32)
            extra["visibility"] = np.concatenate(viss)
        if use_kp:
            extra["keypoints"] = np.concatenate(kps)
        np.savez_compressed(
            os.path.join(out_dir, f"bpbss_{name}.npz"),
            ltwh=np.concatenate(ltwhs), conf=np.concatenate(confs), det_ids=np.concatenate(ids_in),
            det_offsets=np.array(in_off, dtype=np.int64), out_offsets=np.array(out_off, dtype=np.int64),
            o_idx=np.array(o_idx, dtype=np.int64), o_track_id=np.array(o_tid, dtype=np.int64),
            o_kf_ltwh=np.array(o_kf).reshape(-1, 4), o_pred_ltwh=np.array(o_pred).reshape(-1, 4),
            o_pred_valid=np.array(o_pred_valid, dtype=np.int64),
            o_matched_name=np.array(o_mname, dtype=np.int64), o_matched_dist=np.array(o_mdist),
            o_hits=np.array(o_hits, dtype=np.int64), o_age=np.array(o_age, dtype=np.int64),
            o_tsu=np.array(o_tsu, dtype=np.int64), o_state=np.array(o_state, dtype=np.int64),
            config=json.dumps(cfg), seed=seed, n_objects=nobj, n_frames=nframes, parts=K, dim=D,
            stream_kwargs=json.dumps(skw), input_sha256=h.hexdigest(), **extra, **blobs)
        print(f"bpbss_{name}: rows_out={out_off[-1]} next_id={model.tracker._next_id}")


def gen_kf8(out_dir):
    """8-state xyah NSA Kalman unit vectors (bpbreid_strong_sort/sort/kalman_filter.py:53-227)."""
    _install_cv2_stub()
    import bpbreid_strong_sort.sort.kalman_filter as KF
    rng = np.random.default_rng(21)
    kf = KF.KalmanFilter()
    meas = np.stack([rng.uniform(100, 1800, 32), rng.uniform(100, 1000, 32),
                     rng.uniform(0.3, 0.6, 32), rng.uniform(80, 300, 32)], 1)
    blobs = {"meas": meas}
    m0, c0, m1, c1, pm, pc, m2, c2, g4, g2 = ([] for _ in range(10))
    confs = rng.uniform(0.3, 1.0, 32)
    cand = np.stack([rng.uniform(100, 1800, 50), rng.uniform(100, 1000, 50),
                     rng.uniform(0.3, 0.6, 50), rng.uniform(80, 300, 50)], 1)
    for i in range(32):
        mean, cov = kf.initiate(meas[i])
        m0.append(mean), c0.append(cov)
        for _ in range(i % 4 + 1):
            mean, cov = kf.predict(mean, cov)
        m1.append(mean), c1.append(cov)
        a, b = kf.project(mean, cov, confs[i])
        pm.append(a), pc.append(b)
        z = meas[i] + np.r_[rng.normal(0, 3, 2), rng.normal(0, 0.01), rng.normal(0, 3)]
        cand[i % 50] = z
        g4.append(kf.gating_distance(mean, cov, cand.copy(), False))
        g2.append(kf.gating_distance(mean, cov, cand.copy(), True))
        mean, cov = kf.update(mean, cov, z, confs[i])
        m2.append(mean), c2.append(cov)
        blobs[f"z{i}"] = z
    blobs.update(conf=confs, cand_last=cand, init_mean=np.stack(m0), init_cov=np.stack(c0),
                 pred_mean=np.stack(m1), pred_cov=np.stack(c1), proj_mean=np.stack(pm), proj_cov=np.stack(pc),
                 upd_mean=np.stack(m2), upd_cov=np.stack(c2), gate4=np.stack(g4), gate2=np.stack(g2))
    # candidates evolve per case (cand[i%50]=z): store per-case snapshots for exact replay
    np.savez_compressed(os.path.join(out_dir, "kf8_cases.npz"), **blobs)
    print("kf8 ok")


def gen_hota(out_dir):
    """HOTA of a tracker output vs synthetic GT through the TrackEval copy vendored by the reference, loaded file by
    file (trackeval/__init__ pulls shapely): plugins/eval/PoseTrack21/posetrack21/posetrack21/trackeval/."""
    import importlib.util
    base = os.path.join(REF, "plugins", "eval", "PoseTrack21", "posetrack21", "posetrack21", "trackeval")
    pkg = types.ModuleType("trackeval"); pkg.__path__ = [base]; sys.modules["trackeval"] = pkg
    mpkg = types.ModuleType("trackeval.metrics"); mpkg.__path__ = [os.path.join(base, "metrics")]; sys.modules["trackeval.metrics"] = mpkg

    def load(name, path):
        spec = importlib.util.spec_from_file_location(name, path)
        m = importlib.util.module_from_spec(spec); sys.modules[name] = m; spec.loader.exec_module(m); return m
    load("trackeval._timing", os.path.join(base, "_timing.py"))
    load("trackeval.utils", os.path.join(base, "utils.py"))
    load("trackeval.metrics._base_metric", os.path.join(base, "metrics", "_base_metric.py"))
    H = load("trackeval.metrics.hota", os.path.join(base, "metrics", "hota.py")).HOTA()
    _install_filterpy_shim()
    import oc_sort.ocsort as ref
    from tracklab_amd.hota import box_iou_matrix
    blobs = {}
    per_seq = []
    for si, (seed, nobj, nfr, skw) in enumerate([(0, 30, 120, {"miss_prob": 0.1}), (1, 12, 80, {"miss_prob": 0.3, "churn_period": 8})]):
        trk = ref.OCSort(**OCSORT_CONFIGS["yaml"]["hyper"])
        gmap, tmap = {}, {}
        gt_ids, tr_ids, sims = [], [], []
        for fr in SyntheticStream(seed, nobj, nfr, **skw):
            inp = torch.from_numpy(fr["dets"]); inp = inp[inp[:, 4] > 0.4]
            out = np.asarray(trk.update(inp, None), dtype=np.float64).reshape(-1, 8)
            g = np.array([gmap.setdefault(int(x), len(gmap)) for x in fr["gt_all_ids"]], dtype=int)
            t = np.array([tmap.setdefault(int(x), len(tmap)) for x in out[:, 4]], dtype=int)
            sim = box_iou_matrix(fr["gt_boxes"], out[:, :4])
            gt_ids.append(g); tr_ids.append(t); sims.append(sim)
            blobs[f"s{si}_f{fr['frame']}_gt_ids"] = fr["gt_all_ids"]; blobs[f"s{si}_f{fr['frame']}_gt_boxes"] = fr["gt_boxes"]
            blobs[f"s{si}_f{fr['frame']}_tr_ids"] = out[:, 4].astype(np.int64); blobs[f"s{si}_f{fr['frame']}_tr_boxes"] = out[:, :4]
        data = {"num_tracker_dets": sum(len(t) for t in tr_ids), "num_gt_dets": sum(len(g) for g in gt_ids),
                "num_gt_ids": len(gmap), "num_tracker_ids": len(tmap), "num_timesteps": nfr,
                "gt_ids": gt_ids, "tracker_ids": tr_ids, "similarity_scores": sims}
        res = H.eval_sequence(data)
        per_seq.append(res)
        blobs[f"s{si}_n_frames"] = np.int64(nfr)
        for k in ("HOTA", "DetA", "AssA", "DetRe", "DetPr", "AssRe", "AssPr", "LocA", "HOTA_TP", "HOTA_FN", "HOTA_FP"):
            blobs[f"s{si}_{k}"] = np.asarray(res[k], dtype=np.float64)
    comb = H.combine_sequences({"a": per_seq[0], "b": per_seq[1]})
    for k in ("HOTA", "DetA", "AssA", "DetRe", "DetPr", "AssRe", "AssPr", "LocA", "HOTA_TP", "HOTA_FN", "HOTA_FP"):
        blobs[f"comb_{k}"] = np.asarray(comb[k], dtype=np.float64)
    np.savez_compressed(os.path.join(out_dir, "hota_cases.npz"), **blobs)
    print("hota ok", float(np.mean(comb["HOTA"])))


def gen_cosine(out_dir):
    """Cosine gallery metric of plain StrongSORT (strong_sort/sort/nn_matching.py), imported file-by-file (the package
    __init__ chain needs ultralytics/torchvision; this module only needs numpy + torch)."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("ss_nn_matching", os.path.join(REF, "plugins", "track", "strong_sort", "sort", "nn_matching.py"))
    nnm = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(nnm)
    rng = np.random.default_rng(17)
    blobs = {}
    for ci, (T, N, D, gmax) in enumerate([(60, 100, 512, 40), (7, 33, 64, 5), (1, 1, 16, 1), (40, 130, 128, 100)]):
        metric = nnm.NearestNeighborDistanceMetric("cosine", 0.2, 100)
        proto = rng.normal(0, 1, (max(T, N), D)).astype(np.float32)
        sizes = rng.integers(1, gmax + 1, T)
        gal, offs = [], [0]
        for t in range(T):
            feats = [(proto[t] + 0.3 * rng.normal(0, 1, D)).astype(np.float32) for _ in range(sizes[t])]
            metric.samples[t] = feats
            gal += feats
            offs.append(offs[-1] + len(feats))
        dets = (proto[:N] + 0.3 * rng.normal(0, 1, (N, D))).astype(np.float32)
        cost = metric.distance(dets, list(range(T)))
        blobs[f"c{ci}_gallery"], blobs[f"c{ci}_offsets"] = np.stack(gal), np.array(offs, dtype=np.int32)
        blobs[f"c{ci}_dets"], blobs[f"c{ci}_cost"] = dets, np.asarray(cost, dtype=np.float64)
    blobs["n_cases"] = np.int64(4)
    np.savez_compressed(os.path.join(out_dir, "cosine_gallery.npz"), **blobs)
    print("cosine ok")


def gen_motion_costs(out_dir):
    """iou / oks of the StrongSORT family (bpbreid_strong_sort/sort/iou_matching.py:7-39, oks_matching.py:30-92), row by row
    as iou_cost / oks_cost call them (:73-76, :124-127)."""
    _install_cv2_stub()
    import bpbreid_strong_sort.sort.iou_matching as ium
    import bpbreid_strong_sort.sort.oks_matching as okm
    rng = np.random.default_rng(23)
    blobs = {}
    for ci, (T, N) in enumerate([(40, 55), (1, 1), (100, 130), (3, 17)]):
        c = rng.uniform(100, 1800, (max(T, N), 2))
        wh = np.stack([rng.uniform(40, 120, max(T, N)), rng.uniform(90, 300, max(T, N))], 1)
        trk = np.concatenate([c[:T] - wh[:T] / 2 + rng.normal(0, 6, (T, 2)), wh[:T] * rng.uniform(0.9, 1.1, (T, 2))], 1)
        det = np.concatenate([c[:N] - wh[:N] / 2, wh[:N]], 1)
        iou_cost = np.stack([1.0 - ium.iou(trk[t], det) for t in range(T)])

        def skeleton(box, jitter):
            l, t, w, h = box
            kp = np.empty((17, 3))
            kp[:, 0] = l + w * rng.uniform(0.1, 0.9, 17) + rng.normal(0, jitter, 17)
            kp[:, 1] = t + h * np.linspace(0.05, 0.95, 17) + rng.normal(0, jitter, 17)
            kp[:, 2] = np.where(rng.uniform(0, 1, 17) < 0.8, rng.uniform(0.3, 1.0, 17), 0.0)
            kp[0, 2] = max(kp[0, 2], 0.5)
            return kp
        tk = np.stack([skeleton(b, 0.0) for b in trk])
        dk = np.stack([skeleton(b, 3.0) for b in det])
        if ci == 0:
            # degenerate skeletons: collinear visible keypoints (axis-aligned area 0 -> 45-degree fallback), a single visible
            # keypoint (scale < 0.1 -> NaN row), all keypoints visible
            tk[0, :, 0] = tk[0, 0, 0]
            tk[1, :, 2] = 0.0; tk[1, 3, 2] = 0.9
            tk[2, :, 2] = 1.0
        oks_cost = np.stack([1.0 - okm.oks(tk[t], dk) for t in range(T)])
        blobs[f"c{ci}_trk_ltwh"], blobs[f"c{ci}_det_ltwh"], blobs[f"c{ci}_iou_cost"] = trk, det, iou_cost
        blobs[f"c{ci}_trk_kps"], blobs[f"c{ci}_det_kps"], blobs[f"c{ci}_oks_cost"] = tk, dk, oks_cost
    blobs["n_cases"] = np.int64(4)
    np.savez_compressed(os.path.join(out_dir, "motion_costs.npz"), **blobs)
    print("motion costs ok")


SS_DEFAULTS = dict(max_dist=0.2, max_iou_dist=0.7, max_age=70, max_unmatched_preds=7, n_init=3, nn_budget=100, mc_lambda=0.995, ema_alpha=0.9)
SS_YAML = dict(ema_alpha=0.8962157769329083, max_age=40, max_dist=0.1594374041012136, max_iou_dist=0.5431835667667874,
               max_unmatched_preds=0, mc_lambda=0.995, n_init=3, nn_budget=100)     # configs/modules/track/strong_sort.yaml
SS_RUNS = [  # name, hyperparams, seed, objects, frames, D, store embeddings, stream kwargs
    ("defaults_s0_n100_d512", SS_DEFAULTS, 0, 100, 60, 512, False, {}),
    ("yaml_s1_n50_d128", SS_YAML, 1, 50, 120, 128, False, {}),
    ("budget5_s2_n20_d32", dict(SS_DEFAULTS, nn_budget=5, max_age=12), 2, 20, 160, 32, True, dict(miss_prob=0.12, churn_period=30)),
    ("ninit1_s3_n20_d32", dict(SS_DEFAULTS, n_init=1, max_unmatched_preds=0, max_age=8, max_dist=0.3), 3, 20, 160, 32, True,
     dict(miss_prob=0.1, churn_period=25)),
    ("lowconf_s4_n30_d64", dict(SS_DEFAULTS, max_iou_dist=0.5), 4, 30, 120, 64, False, dict(miss_prob=0.05, churn_period=40, low_conf_frac=0.3)),
    ("crowded_s5_n110_d64", dict(SS_YAML, nn_budget=30), 5, 110, 80, 64, False, dict(miss_prob=0.1, churn_period=15)),                              # r03
]


def _import_plain_strong_sort():
    """plugins/track/strong_sort/strong_sort.py with its heavy imports stubbed: gdown / torchvision / ultralytics are only used by
    the ReID model loader (never constructed here); ultralytics.utils.ops.xyxy2xywh is the one function update() calls and is
    restated from its published definition (centre = mean of corners, size = difference)."""
    import types
    _install_cv2_stub()

    def stub(name, **attrs):
        m = types.ModuleType(name)
        m.__dict__.update(attrs)
        sys.modules[name] = m
        return m

    def xyxy2xywh(x):
        y = np.empty_like(x)
        y[..., 0] = (x[..., 0] + x[..., 2]) / 2
        y[..., 1] = (x[..., 1] + x[..., 3]) / 2
        y[..., 2] = x[..., 2] - x[..., 0]
        y[..., 3] = x[..., 3] - x[..., 1]
        return y
    if "gdown" not in sys.modules:
        stub("gdown")
    if "torchvision" not in sys.modules:
        tv = stub("torchvision")
        tv.transforms = stub("torchvision.transforms")
    for name, attrs in (("ultralytics", {}), ("ultralytics.utils", dict(LOGGER=None)), ("ultralytics.utils.ops", {}),
                        ("ultralytics.utils.checks", dict(check_requirements=None, check_version=None))):
        m = sys.modules.get(name) or stub(name)
        m.__dict__.update(attrs)
        m.__path__ = getattr(m, "__path__", [])             # a package, so that submodule imports resolve through sys.modules
    if not hasattr(sys.modules["ultralytics.utils.ops"], "xyxy2xywh"):
        sys.modules["ultralytics.utils.ops"].xyxy2xywh = xyxy2xywh
    import strong_sort.strong_sort as ss
    from strong_sort.sort.nn_matching import NearestNeighborDistanceMetric
    from strong_sort.sort.tracker import Tracker
    return ss, NearestNeighborDistanceMetric, Tracker


def gen_ssort(out_dir):
    """Plain StrongSORT (plugins/track/strong_sort): StrongSORT.update (strong_sort.py:41-84) run as is, with _get_features
    (the OSNet forward on image crops) replaced by the synthetic stream's embeddings."""
    ss, Metric, Tracker = _import_plain_strong_sort()
    for name, hp, seed, nobj, nframes, D, store, skw in SS_RUNS:
        model = object.__new__(ss.StrongSORT)            # __init__ would load ReID weights (strong_sort.py:33)
        model.max_dist = hp["max_dist"]
        model.tracker = Tracker(Metric("cosine", hp["max_dist"], hp["nn_budget"]), max_iou_dist=hp["max_iou_dist"], max_age=hp["max_age"],
                                n_init=hp["n_init"], max_unmatched_preds=hp["max_unmatched_preds"], mc_lambda=hp["mc_lambda"],
                                ema_alpha=hp["ema_alpha"])     # strong_sort.py:36-39
        stream = SyntheticStream(seed, nobj, nframes, parts=1, dim=D, with_embeddings=True, **skw)
        frame_img = np.zeros((1080, 1920, 3), dtype=np.uint8)
        in_off, out_off = [0], [0]
        dets_all, embs, rows = [], [], []
        h = hashlib.sha256()
        blobs = {}
        for fr in stream:
            dets = fr["dets"]
            emb = fr["embeddings"][:, 0, :].astype(np.float32)
            if fr["frame"] % 37 == 11:
                dets, emb = dets[:0], emb[:0]
            h.update(np.ascontiguousarray(dets).tobytes()); h.update(np.ascontiguousarray(emb).tobytes())
            dets_all.append(dets)
            if store:
                embs.append(emb)
            in_off.append(in_off[-1] + len(dets))
            n_out = 0
            if len(dets) > 0:                             # wrapper: process() returns [] on an empty frame (strong_sort_api.py:68-69)
                feats = torch.from_numpy(emb.copy())
                model._get_features = lambda xywhs, img, feats=feats: feats
                out = model.update(torch.from_numpy(dets.copy()), frame_img)
                for r in out:
                    rows.append([float(r[0]), float(r[1]), float(r[2]), float(r[3]), float(r[4]), float(r[5]), float(r[6]), float(r[8])])
                    n_out += 1
                f = fr["frame"]
                if f in (2, 9, 30, 59, 100, 150) and len(model.tracker.tracks):
                    tr = model.tracker.tracks
                    blobs[f"f{f}_track_ids"] = np.array([t.track_id for t in tr], dtype=np.int64)
                    blobs[f"f{f}_mean"] = np.stack([t.mean for t in tr])
                    blobs[f"f{f}_cov"] = np.stack([t.covariance for t in tr])
                    blobs[f"f{f}_feat"] = np.stack([t.features[-1] for t in tr]).astype(np.float32)
                    blobs[f"f{f}_state"] = np.array([[t.hits, t.age, t.time_since_update, t.state, t.updates_wo_assignment] for t in tr], dtype=np.int64)
                    blobs[f"f{f}_gallery"] = np.array([len(model.tracker.metric.samples.get(t.track_id, [])) for t in tr], dtype=np.int64)
            out_off.append(out_off[-1] + n_out)
        extra = {"embeddings": np.concatenate(embs).astype(np.float32)} if store else {}
        np.savez_compressed(
            os.path.join(out_dir, f"ssort_{name}.npz"), dets=np.concatenate(dets_all), det_offsets=np.array(in_off, dtype=np.int64),
            out_offsets=np.array(out_off, dtype=np.int64), rows=np.array(rows, dtype=np.float64).reshape(-1, 8),
            config=json.dumps(hp), seed=seed, n_objects=nobj, n_frames=nframes, dim=D, stream_kwargs=json.dumps(skw),
            input_sha256=h.hexdigest(), **extra, **blobs)
        print(f"ssort_{name}: rows_out={out_off[-1]} next_id={model.tracker._next_id}")


def gen_pil_preprocess(out_dir):
    """Plain StrongSORT's ReID input: crop ori_img[y1:y2, x1:x2] of the int-truncated, clipped box
    (strong_sort.py:102-108, :135-141), then ReIDDetectMultiBackend._preprocess (reid_multibackend.py:44-52, :184-195):
    ToPILImage -> Resize((256, 128)) -> ToTensor -> Normalize(ImageNet). torchvision is not installed; its three transforms
    are thin wrappers whose arithmetic is Pillow's Image.resize(BILINEAR) [third-party, Pillow 12.2.0 here] followed by
    uint8 -> float32 .div(255) and .sub_(mean).div_(std) on torch tensors, which is what runs below."""
    from PIL import Image
    rng = np.random.default_rng(31)
    H, W = 540, 960
    yy, xx = np.mgrid[0:H, 0:W]
    img = np.stack([(xx * 255 // W), (yy * 255 // H), ((xx + yy) % 256)], axis=2).astype(np.int64)
    img = np.clip(img + rng.integers(-40, 40, img.shape), 0, 255).astype(np.uint8)
    for _ in range(30):                                      # flat rectangles like the synthetic renderer
        x, y, w, h = rng.integers(0, W - 80), rng.integers(0, H - 160), rng.integers(20, 80), rng.integers(40, 160)
        img[y:y + h, x:x + w] = rng.integers(0, 255, 3)
    boxes = np.array([
        [100.3, 50.7, 161.9, 203.2],      # typical person: upscale in both directions
        [400.0, 100.0, 528.0, 356.0],     # exactly 128 x 256: Pillow still resamples only if the size differs -> copy
        [10.2, 20.9, 330.7, 520.1],       # larger than the target: antialiased downscale in both directions
        [600.5, 30.5, 900.4, 180.9],      # wide and short: downscale x, upscale y
        [-15.0, -8.0, 40.6, 99.0],        # clipped at the top-left
        [930.2, 400.0, 975.0, 560.0],     # clipped at the bottom-right
        [300.0, 300.0, 303.9, 420.0],     # 3 px wide
        [500.0, 200.0, 590.0, 204.2],     # 4 px tall
        [700.1, 250.2, 701.9, 252.8],     # 1 x 2 px
        [50.9, 260.1, 179.2, 389.9],      # 128 px wide, shorter than 256
    ])

    def xyxy_int(b):                                          # xyxy2xywh then _xywh_to_xyxy (strong_sort.py:102-108)
        x, y, w, h = (b[0] + b[2]) / 2, (b[1] + b[3]) / 2, b[2] - b[0], b[3] - b[1]
        return max(int(x - w / 2), 0), max(int(y - h / 2), 0), min(int(x + w / 2), W - 1), min(int(y + h / 2), H - 1)
    mean = torch.tensor([0.485, 0.456, 0.406]).view(3, 1, 1)
    std = torch.tensor([0.229, 0.224, 0.225]).view(3, 1, 1)
    resized, ints, tensors = [], [], {}
    for i, b in enumerate(boxes):
        x1, y1, x2, y2 = xyxy_int(b)
        crop = img[y1:y2, x1:x2]
        pil = Image.fromarray(crop)                           # ToPILImage on an (H, W, 3) uint8 ndarray
        r = np.asarray(pil.resize((128, 256), Image.BILINEAR))        # Resize((256, 128)) = (h, w); PIL takes (w, h)
        resized.append(r); ints.append([x1, y1, x2, y2])
        if i in (0, 2):
            t = torch.from_numpy(r.copy()).permute(2, 0, 1).contiguous().to(torch.float32).div(255)      # ToTensor
            tensors[f"norm{i}"] = t.sub_(mean).div_(std).numpy()                                          # Normalize
    lut = torch.arange(256, dtype=torch.uint8).view(1, 256, 1).repeat(3, 1, 1).to(torch.float32).div(255).sub_(mean).div_(std)
    np.savez_compressed(os.path.join(out_dir, "pil_preprocess.npz"), image=img, boxes=boxes, boxes_int=np.array(ints, dtype=np.int64),
                        resized=np.stack(resized), norm_lut=lut.numpy()[:, :, 0], pillow=np.array(Image.__version__), **tensors)
    print("pil preprocess ok", len(boxes))


def pil_sweep_image(w_img, H=720):
    """The sweep's frame, from arithmetic only (no random generator: the tests rebuild it instead of storing 2.7 MB): gradients, a hash-like
    texture of +-6 and flat rectangles."""
    yy, xx = np.mgrid[0:H, 0:w_img].astype(np.int64)
    tex = ((xx * 1103515245 + yy * 12345 + xx * yy * 7) >> 3) % 13 - 6
    img = np.stack([(xx * 255 // w_img) + tex, (yy * 255 // H) - tex, ((xx * 3 + yy * 5) % 256) + tex // 2], axis=2)
    for k in range(60):
        x, y = (k * 197) % (w_img - 90), (k * 113) % (H - 170)
        w, h = 10 + (k * 31) % 70, 10 + (k * 53) % 150
        img[y:y + h, x:x + w] = [(k * 37) % 256, (k * 91) % 256, (k * 151) % 256]
    return np.clip(img, 0, 255).astype(np.uint8)


PIL_SWEEP_SIZES = [(20, 40), (64, 128), (100, 250), (127, 255), (128, 256), (129, 257), (140, 300), (150, 306), (153, 307), (154, 308), (155, 310), (160, 400),
                   (160, 512), (161, 513), (100, 600), (200, 300), (3, 500), (128, 10), (97, 193), (50, 511), (159, 2), (2, 2), (120, 270), (80, 330), (33, 77),
                   (145, 290), (111, 222), (90, 180), (60, 500), (158, 316)]


def gen_pil_sweep(out_dir):
    """r03: a sweep of crop sizes through Pillow itself for the specialised paths of pil_wave_kernel (tap counts 2 / 3 / 5 per axis, 4- and 2-row
    mini-bands around a vertical scale of 1.2, staged vs direct crops around 160 px width and 2x height, tiny and clipped crops, row pitches that
    are / are not multiples of 16 bytes): same chain as gen_pil_preprocess (crop of the int-truncated clipped box -> Image.resize((128, 256),
    BILINEAR)). The fixture holds, per crop, the SHA-256 of Pillow's resized uint8 array and every 8th of its rows; the frames are rebuilt by the tests."""
    import hashlib

    from PIL import Image
    out = {}
    for tag, w_img in (("a", 1280), ("b", 1283)):
        H = 720
        img = pil_sweep_image(w_img, H)
        boxes, ints, shas, rows = [], [], [], []
        for k, (cw, ch) in enumerate(PIL_SWEEP_SIZES if tag == "a" else PIL_SWEEP_SIZES[::3]):
            x1, y1 = (k * 211) % max(1, w_img - cw - 1), (k * 97) % max(1, H - ch - 1)
            if k % 7 == 6:                                     # clipped at the bottom-right corner: the last rows of the frame
                x1, y1 = w_img - 1 - cw // 2, H - 1 - ch // 2
            b = np.array([x1 + 0.3, y1 + 0.6, x1 + cw + 0.4, y1 + ch + 0.2])
            x, y, w, h = (b[0] + b[2]) / 2, (b[1] + b[3]) / 2, b[2] - b[0], b[3] - b[1]
            ix1, iy1, ix2, iy2 = max(int(x - w / 2), 0), max(int(y - h / 2), 0), min(int(x + w / 2), w_img - 1), min(int(y + h / 2), H - 1)
            crop = img[iy1:iy2, ix1:ix2]
            if crop.shape[0] == 0 or crop.shape[1] == 0:
                continue
            r = np.ascontiguousarray(np.asarray(Image.fromarray(np.ascontiguousarray(crop)).resize((128, 256), Image.BILINEAR)))
            boxes.append(b); ints.append([ix1, iy1, ix2, iy2]); shas.append(hashlib.sha256(r.tobytes()).hexdigest()); rows.append(r[::8])
        out[f"boxes_{tag}"] = np.array(boxes); out[f"boxes_int_{tag}"] = np.array(ints, dtype=np.int64)
        out[f"sha_{tag}"] = np.array(shas); out[f"rows_{tag}"] = np.stack(rows); out[f"width_{tag}"] = np.int64(w_img)
    mean = torch.tensor([0.485, 0.456, 0.406]).view(3, 1, 1)
    std = torch.tensor([0.229, 0.224, 0.225]).view(3, 1, 1)
    lut = torch.arange(256, dtype=torch.uint8).view(1, 256, 1).repeat(3, 1, 1).to(torch.float32).div(255).sub_(mean).div_(std)
    path = os.path.join(out_dir, "pil_sweep.npz")
    np.savez_compressed(path, norm_lut=lut.numpy()[:, :, 0], pillow=np.array(Image.__version__), **out)
    print("pil sweep ok", len(out["boxes_a"]), len(out["boxes_b"]), os.path.getsize(path))


BT_RUNS = [  # name, hyperparams, seed, objects, frames, stream kwargs
    ("yaml_s0_n100", dict(track_thresh=0.6, track_buffer=30, match_thresh=0.8, frame_rate=30), 0, 100, 80, dict(low_conf_frac=0.2)),
    ("defaults_s1_n50", dict(track_thresh=0.45, match_thresh=0.8, track_buffer=25, frame_rate=30), 1, 50, 150, dict(miss_prob=0.08, churn_period=40, low_conf_frac=0.3)),
    ("short_buffer_s2_n20", dict(track_thresh=0.5, match_thresh=0.7, track_buffer=5, frame_rate=30), 2, 20, 200, dict(miss_prob=0.15, churn_period=25, low_conf_frac=0.3)),
    ("fps15_s3_n30", dict(track_thresh=0.6, match_thresh=0.9, track_buffer=30, frame_rate=15), 3, 30, 150, dict(miss_prob=0.1, churn_period=30, low_conf_frac=0.4)),
    ("crowded_s4_n120", dict(track_thresh=0.6, track_buffer=30, match_thresh=0.8, frame_rate=30), 4, 120, 100, dict(miss_prob=0.1, churn_period=15, low_conf_frac=0.25)),      # r03
]


def _import_byte_track():
    """plugins/track/byte_track with `lap` (not installed) shimmed: lapjv(cost, extend_cost=True, cost_limit=L) = optimum of the
    documented (nr+nc)^2 embedding (padding L/2, zero lower-right block), solved with scipy -- the matched pairs are the unique
    optimum whenever no two real costs tie, whatever solver finds it -- and ultralytics' box conversions restated."""
    import types
    from scipy.optimize import linear_sum_assignment

    def lapjv(cost, extend_cost=False, cost_limit=np.inf, return_cost=True):
        cost = np.ascontiguousarray(cost, dtype=np.double)
        nr, nc = cost.shape
        n = nr + nc
        ext = np.full((n, n), cost_limit / 2.0)
        ext[nr:, nc:] = 0
        ext[:nr, :nc] = cost
        r, c = linear_sum_assignment(ext)
        x = np.full(nr, -1, dtype=int); y = np.full(nc, -1, dtype=int)
        for i, j in zip(r, c):
            if i < nr and j < nc:
                x[i] = j; y[j] = i
        return float(cost[x >= 0, x[x >= 0]].sum()), x, y

    def xyxy2xywh(x):
        y = np.empty_like(x)
        y[..., 0] = (x[..., 0] + x[..., 2]) / 2; y[..., 1] = (x[..., 1] + x[..., 3]) / 2
        y[..., 2] = x[..., 2] - x[..., 0]; y[..., 3] = x[..., 3] - x[..., 1]
        return y

    def xywh2xyxy(x):
        y = np.empty_like(x)
        xy, wh = x[..., :2], x[..., 2:] / 2
        y[..., :2] = xy - wh; y[..., 2:] = xy + wh
        return y
    m = types.ModuleType("lap"); m.lapjv = lapjv; sys.modules["lap"] = m
    for name in ("ultralytics", "ultralytics.utils"):
        if name not in sys.modules:
            sys.modules[name] = types.ModuleType(name)
        sys.modules[name].__path__ = getattr(sys.modules[name], "__path__", [])
    ops = sys.modules.get("ultralytics.utils.ops") or types.ModuleType("ultralytics.utils.ops")
    ops.xyxy2xywh, ops.xywh2xyxy = xyxy2xywh, xywh2xyxy
    sys.modules["ultralytics.utils.ops"] = ops
    import byte_track.byte_tracker as bt
    from byte_track.basetrack import BaseTrack
    return bt, BaseTrack


def gen_bytetrack(out_dir):
    """ByteTrack (plugins/track/byte_track): BYTETracker.update run as is behind the wrapper's confidence filter
    (wrappers/track/byte_track_api.py:57-60)."""
    bt, BaseTrack = _import_byte_track()
    for name, hp, seed, nobj, nframes, skw in BT_RUNS:
        BaseTrack._count = 0                               # class-level id counter: our handles restart at 1 per stream
        model = bt.BYTETracker(**hp)
        stream = SyntheticStream(seed, nobj, nframes, **skw)
        in_off, out_off, dets_all, rows = [0], [0], [], []
        blobs = {}
        for fr in stream:
            dets = fr["dets"]
            if fr["frame"] % 43 == 17:
                dets = dets[:0]
            dets_all.append(dets)
            in_off.append(in_off[-1] + len(dets))
            n_out = 0
            if len(dets) > 0:
                inputs = torch.from_numpy(dets.copy())
                inputs = inputs[inputs[:, 4] > 0.4]
                out = model.update(inputs, None)
                for r in out:
                    rows.append([float(v) for v in r])
                    n_out += 1
                f = fr["frame"]
                if f in (0, 1, 2, 10, 40, 79, 120, 199):
                    for lname, lst in (("trk", model.tracked_stracks), ("lost", model.lost_stracks)):
                        blobs[f"f{f}_{lname}_ids"] = np.array([t.track_id for t in lst], dtype=np.int64)
                        blobs[f"f{f}_{lname}_mean"] = np.array([np.asarray(t.mean, dtype=np.float64) for t in lst]).reshape(-1, 8)
                        blobs[f"f{f}_{lname}_cov"] = np.array([np.asarray(t.covariance, dtype=np.float64) for t in lst]).reshape(-1, 8, 8)
                        blobs[f"f{f}_{lname}_state"] = np.array([[t.state, int(t.is_activated), t.frame_id, t.start_frame, t.tracklet_len]
                                                               for t in lst], dtype=np.int64).reshape(-1, 5)
            out_off.append(out_off[-1] + n_out)
        np.savez_compressed(
            os.path.join(out_dir, f"bytetrack_{name}.npz"), dets=np.concatenate(dets_all), det_offsets=np.array(in_off, dtype=np.int64),
            out_offsets=np.array(out_off, dtype=np.int64), rows=np.array(rows, dtype=np.float64).reshape(-1, 8), config=json.dumps(hp),
            seed=seed, n_objects=nobj, n_frames=nframes, stream_kwargs=json.dumps(skw), min_confidence=0.4, **blobs)
        print(f"bytetrack_{name}: rows_out={out_off[-1]} next_id={BaseTrack._count + 1} lost={len(model.lost_stracks)} removed={len(model.removed_stracks)}")


BOT_DEFAULTS = dict(track_high_thresh=0.45, new_track_thresh=0.6, track_buffer=30, match_thresh=0.8, proximity_thresh=0.5,
                    appearance_thresh=0.25, frame_rate=30, lambda_=0.985)
BOT_YAML = dict(appearance_thresh=0.4818211117541298, frame_rate=30, lambda_=0.9896143462366406, match_thresh=0.22734550911325851,
                new_track_thresh=0.21144301345190655, proximity_thresh=0.5945380911899254, track_buffer=60,
                track_high_thresh=0.33824964456239337)          # configs/modules/track/bot_sort.yaml (cmc_method replaced by none)
BOT_RUNS = [  # name, hyperparams, seed, objects, frames, D, stream kwargs
    ("defaults_s0_n60_d128", BOT_DEFAULTS, 0, 60, 100, 128, dict(low_conf_frac=0.2, miss_prob=0.05)),
    ("yaml_s1_n40_d64", BOT_YAML, 1, 40, 150, 64, dict(low_conf_frac=0.3, miss_prob=0.1, churn_period=40)),
    ("short_s2_n20_d32", dict(BOT_DEFAULTS, track_buffer=6, match_thresh=0.6, appearance_thresh=0.4), 2, 20, 200, 32,
     dict(low_conf_frac=0.3, miss_prob=0.15, churn_period=25)),
    ("classes_s3_n25_d32", dict(BOT_DEFAULTS, new_track_thresh=0.5), 3, 25, 120, 32, dict(low_conf_frac=0.25, miss_prob=0.1, churn_period=30)),
    ("crowded_s4_n110_d64", BOT_YAML, 4, 110, 100, 64, dict(low_conf_frac=0.2, miss_prob=0.1, churn_period=15)),                                 # r03
]


def gen_botsort(out_dir):
    """BoT-SORT (plugins/track/bot_sort): BoTSORT.update run as is with cmc_method 'none' (the camera-motion estimators are cv2),
    `lap` shimmed like for ByteTrack, the ReID forward (_get_features) replaced by synthetic embeddings."""
    _import_byte_track()                                  # installs the lap / ultralytics shims
    _import_plain_strong_sort()                           # installs the gdown / torchvision stubs the ReID loader imports
    import bot_sort.bot_sort as bs
    from bot_sort.basetrack import BaseTrack
    from bot_sort.gmc import GMC
    from bot_sort.kalman_filter import KalmanFilter
    for name, hp, seed, nobj, nframes, D, skw in BOT_RUNS:
        model = object.__new__(bs.BoTSORT)                # __init__ loads ReID weights (bot_sort.py:237-270)
        model.tracked_stracks, model.lost_stracks, model.removed_stracks = [], [], []
        BaseTrack.clear_count()
        model.frame_id = 0
        model.lambda_ = hp["lambda_"]; model.track_high_thresh = hp["track_high_thresh"]; model.new_track_thresh = hp["new_track_thresh"]
        model.buffer_size = int(hp["frame_rate"] / 30.0 * hp["track_buffer"]); model.max_time_lost = model.buffer_size
        model.kalman_filter = KalmanFilter()
        model.proximity_thresh = hp["proximity_thresh"]; model.appearance_thresh = hp["appearance_thresh"]; model.match_thresh = hp["match_thresh"]
        model.gmc = GMC(method="none", verbose=[None, False])
        stream = SyntheticStream(seed, nobj, nframes, parts=1, dim=D, with_embeddings=True, **skw)
        frame_img = np.zeros((1080, 1920, 3), dtype=np.uint8)
        rng_cls = np.random.default_rng(500 + seed)
        in_off, out_off, dets_all, rows = [0], [0], [], []
        blobs = {}
        for fr in stream:
            dets = fr["dets"].copy()
            emb = fr["embeddings"][:, 0, :].astype(np.float32)
            if name.startswith("classes"):
                dets[:, 5] = rng_cls.integers(0, 3, len(dets))            # noisy class labels: exercises the cls_hist vote
            if fr["frame"] % 41 == 13:
                dets, emb = dets[:0], emb[:0]
            dets_all.append(dets)
            in_off.append(in_off[-1] + len(dets))
            n_out = 0
            if len(dets) > 0:
                keep = dets[:, 4] > 0.4                                   # wrapper filter (bot_sort_api.py:67)
                d_in, e_in = dets[keep], emb[keep]
                hi = d_in[:, 4] > hp["track_high_thresh"]
                feats = torch.from_numpy(e_in[hi].copy())
                model._get_features = lambda xywh, img, feats=feats: feats
                out = model.update(torch.from_numpy(d_in.copy()), frame_img)
                for r in out:
                    rows.append([float(v) for v in r]); n_out += 1
                f = fr["frame"]
                if f in (0, 1, 2, 10, 40, 99, 149, 199):
                    for lname, lst in (("trk", model.tracked_stracks), ("lost", model.lost_stracks)):
                        blobs[f"f{f}_{lname}_ids"] = np.array([t.track_id for t in lst], dtype=np.int64)
                        blobs[f"f{f}_{lname}_mean"] = np.array([np.asarray(t.mean, dtype=np.float64) for t in lst]).reshape(-1, 8)
                        blobs[f"f{f}_{lname}_cov"] = np.array([np.asarray(t.covariance, dtype=np.float64) for t in lst]).reshape(-1, 8, 8)
                        blobs[f"f{f}_{lname}_state"] = np.array([[t.state, int(t.is_activated), t.frame_id, t.start_frame, t.tracklet_len]
                                                               for t in lst], dtype=np.int64).reshape(-1, 5)
                        blobs[f"f{f}_{lname}_feat"] = np.array([np.asarray(t.smooth_feat, dtype=np.float32) for t in lst]).reshape(-1, D)
            out_off.append(out_off[-1] + n_out)
        np.savez_compressed(
            os.path.join(out_dir, f"botsort_{name}.npz"), dets=np.concatenate(dets_all), det_offsets=np.array(in_off, dtype=np.int64),
            out_offsets=np.array(out_off, dtype=np.int64), rows=np.array(rows, dtype=np.float64).reshape(-1, 8), config=json.dumps(hp),
            seed=seed, n_objects=nobj, n_frames=nframes, dim=D, stream_kwargs=json.dumps(skw), min_confidence=0.4,
            noisy_classes=int(name.startswith("classes")), **blobs)
        print(f"botsort_{name}: rows_out={out_off[-1]} next_id={BaseTrack._count + 1} lost={len(model.lost_stracks)}")


DOC_YAML = dict(det_thresh=0, max_age=50, min_hits=1, iou_threshold=0.22136877277096445, delta_t=1, asso_func="giou",
                inertia=0.3941737016672115, w_association_emb=0.75, alpha_fixed_emb=0.95, aw_param=0.5, embedding_off=False,
                cmc_off=True, aw_off=False, new_kf_off=False)     # configs/modules/track/deep_oc_sort.yaml (cmc_off: cv2 optical flow)
DOC_DEFAULTS = dict(det_thresh=0.3, max_age=30, min_hits=3, iou_threshold=0.3, delta_t=3, asso_func="iou", inertia=0.2,
                    w_association_emb=0.75, alpha_fixed_emb=0.95, aw_param=0.5, embedding_off=False, cmc_off=True, aw_off=False,
                    new_kf_off=False)
DOC_RUNS = [  # name, hyperparams, seed, objects, frames, D, normalise the detector embeddings, stream kwargs
    ("yaml_s0_n60_d128", DOC_YAML, 0, 60, 100, 128, True, dict(miss_prob=0.05)),
    ("defaults_s1_n40_d64", DOC_DEFAULTS, 1, 40, 150, 64, True, dict(low_conf_frac=0.2, miss_prob=0.15, churn_period=40)),
    ("rawemb_awoff_s2_n20_d32", dict(DOC_YAML, aw_off=True, max_age=8, asso_func="diou", delta_t=2), 2, 20, 200, 32, False,
     dict(miss_prob=0.2, churn_period=25)),
    ("classes_s3_n25_d32", dict(DOC_YAML, asso_func="ciou", w_association_emb=0.4, aw_param=0.7, alpha_fixed_emb=0.8), 3, 25, 120, 32, True,
     dict(miss_prob=0.1, churn_period=30)),
    ("crowded_s4_n110_d64", DOC_YAML, 4, 110, 100, 64, True, dict(miss_prob=0.1, churn_period=15)),                                                 # r03
]


def gen_deepocsort(out_dir):
    """Deep-OC-SORT (plugins/track/deep_oc_sort): OCSort.update run as is with cmc_off (CMCComputer is cv2 optical flow), the ReID
    forward (_get_features) replaced by synthetic float32 torch embeddings (the reference mixes torch tensors and numpy scalars:
    the dtype trail of the track embeddings comes from running it), scipy's linear_sum_assignment (`lap` is not installed in
    the reference's fallback path: association.py:202-212)."""
    _install_filterpy_shim()
    _import_plain_strong_sort()                           # cv2 / gdown / torchvision / ultralytics stubs for the module imports
    saved_lap = sys.modules.get("lap", "absent")
    sys.modules["lap"] = None                             # `import lap` raises ImportError -> scipy fallback
    import deep_oc_sort.ocsort as doc
    try:
        for name, hp, seed, nobj, nframes, D, normed, skw in DOC_RUNS:
            model = object.__new__(doc.OCSort)            # __init__ loads ReID weights and builds the cv2 CMC (ocsort.py:386-391)
            model.max_age, model.min_hits, model.iou_threshold = hp["max_age"], hp["min_hits"], hp["iou_threshold"]
            model.trackers, model.frame_count, model.det_thresh, model.delta_t = [], 0, hp["det_thresh"], hp["delta_t"]
            model.asso_func, model.inertia = doc.ASSO_FUNCS[hp["asso_func"]], hp["inertia"]
            model.w_association_emb, model.alpha_fixed_emb, model.aw_param = hp["w_association_emb"], hp["alpha_fixed_emb"], hp["aw_param"]
            doc.KalmanBoxTracker.count = 0
            model.embedding_off, model.cmc_off, model.aw_off, model.new_kf_off = hp["embedding_off"], hp["cmc_off"], hp["aw_off"], hp["new_kf_off"]
            stream = SyntheticStream(seed, nobj, nframes, parts=1, dim=D, with_embeddings=True, **skw)
            frame_img = np.zeros((1080, 1920, 3), dtype=np.uint8)
            rng_cls = np.random.default_rng(700 + seed)
            in_off, out_off, dets_all, embs_all, rows = [0], [0], [], [], []
            blobs = {}
            for fr in stream:
                dets = fr["dets"].copy()
                emb = fr["embeddings"][:, 0, :].astype(np.float32)
                if normed:
                    emb = emb / np.linalg.norm(emb, axis=1, keepdims=True)
                if name.startswith("classes"):
                    dets[:, 5] = rng_cls.integers(0, 3, len(dets))        # class 0 zeroes the angle cost ("scores" is the class column)
                if fr["frame"] % 43 == 17:
                    dets, emb = dets[:0], emb[:0]
                dets_all.append(dets); embs_all.append(emb)
                in_off.append(in_off[-1] + len(dets))
                n_out = 0
                if len(dets) > 0:
                    keep = dets[:, 4] > 0.4                               # wrapper filter (deep_oc_sort_api.py:62)
                    d_in, e_in = dets[keep], emb[keep]
                    thr = d_in[:, 4] > hp["det_thresh"]
                    feats = torch.from_numpy(e_in[thr].copy())
                    model._get_features = lambda xyxy, img, feats=feats: feats
                    out = model.update(torch.from_numpy(d_in.copy()), frame_img)
                    for r in np.asarray(out, dtype=np.float64).reshape(-1, 8):
                        rows.append(list(r)); n_out += 1
                    f = fr["frame"]
                    if f in (0, 1, 2, 3, 10, 40, 99, 149, 199):
                        T = model.trackers
                        blobs[f"f{f}_ids"] = np.array([t.id for t in T], dtype=np.int64)
                        blobs[f"f{f}_x"] = np.array([t.kf.x[:, 0] for t in T], dtype=np.float64).reshape(-1, 8)
                        blobs[f"f{f}_P"] = np.array([t.kf.P for t in T], dtype=np.float64).reshape(-1, 8, 8)
                        blobs[f"f{f}_emb"] = np.array([np.asarray(t.emb, dtype=np.float64) for t in T]).reshape(-1, D)
                        blobs[f"f{f}_emb_f64"] = np.array([int(np.asarray(t.emb).dtype == np.float64) for t in T], dtype=np.int64)
                        blobs[f"f{f}_state"] = np.array([[t.time_since_update, t.hits, t.hit_streak, t.age, int(t.frozen), int(t.kf.observed)]
                                                         for t in T], dtype=np.int64).reshape(-1, 6)
                        blobs[f"f{f}_vel"] = np.array([t.velocity if t.velocity is not None else (0, 0) for t in T], dtype=np.float64).reshape(-1, 2)
                        blobs[f"f{f}_last"] = np.array([t.last_observation for t in T], dtype=np.float64).reshape(-1, 5)
                out_off.append(out_off[-1] + n_out)
            np.savez_compressed(
                os.path.join(out_dir, f"deepocsort_{name}.npz"), dets=np.concatenate(dets_all), embeddings=np.concatenate(embs_all),
                det_offsets=np.array(in_off, dtype=np.int64), out_offsets=np.array(out_off, dtype=np.int64),
                rows=np.array(rows, dtype=np.float64).reshape(-1, 8), config=json.dumps(hp), seed=seed, n_objects=nobj, n_frames=nframes, dim=D,
                stream_kwargs=json.dumps(skw), min_confidence=0.4, **blobs)
            print(f"deepocsort_{name}: rows_out={out_off[-1]} next_id={doc.KalmanBoxTracker.count} live={len(model.trackers)} "
                  f"f64_embs={sum(int(np.asarray(t.emb).dtype == np.float64) for t in model.trackers)}")
    finally:
        if saved_lap == "absent":
            sys.modules.pop("lap", None)
        else:
            sys.modules["lap"] = saved_lap


CLEARMOT_METRICS = ["num_frames", "num_matches", "num_switches", "num_transfer", "num_ascend", "num_migrate", "num_false_positives", "num_misses",
                    "num_objects", "num_predictions", "num_unique_objects", "mostly_tracked", "partially_tracked", "mostly_lost",
                    "num_fragmentations", "num_detections", "idtp", "idfp", "idfn", "motp", "mota", "precision", "recall", "idp", "idr", "idf1"]


def gen_clearmot(out_dir):
    """CLEAR-MOT / ID measures: the py-motmetrics copy vendored by the reference (plugins/eval/PoseTrack21/posetrack21_mot) run on
    synthetic sequences (ground truth vs. jittered, dropped, id-swapped and spurious hypotheses), scipy solver (the `lap` solver its
    evaluate_mot.py selects is not installed), per sequence and overall (compute_many(generate_overall=True))."""
    import types
    sys.path.insert(0, os.path.join(REF, "plugins", "eval", "PoseTrack21", "posetrack21_mot", "posetrack21_mot"))
    if not hasattr(np, "asfarray"):
        np.asfarray = lambda a, dtype=float: np.asarray(a, dtype=dtype)          # numpy 2 removed it (distances.py:122)
    sys.modules.setdefault("xmltodict", types.ModuleType("xmltodict"))           # motmetrics/io.py imports it for a loader not used here
    import motmetrics as mm
    mm.lap.default_solver = "scipy"
    rng = np.random.default_rng(77)
    seqs, accs = [], []
    for s, (nobj, nframes, jitter, drop, swap, spurious) in enumerate([(8, 60, 2.0, 0.05, 0.02, 0.3), (20, 80, 6.0, 0.2, 0.05, 1.0),
                                                                       (5, 40, 1.0, 0.0, 0.0, 0.0), (12, 50, 15.0, 0.4, 0.1, 2.0)]):
        stream = SyntheticStream(300 + s, nobj, nframes, miss_prob=0.0)
        acc = mm.MOTAccumulator(auto_id=True)
        perm = {}
        frames = []
        for fr in stream:
            gt_ids = fr["gt_all_ids"].astype(np.int64)
            gb = fr["gt_boxes"]
            gt = np.column_stack([gb[:, 0], gb[:, 1], gb[:, 2] - gb[:, 0], gb[:, 3] - gb[:, 1]])
            keep = rng.random(len(gt)) >= drop
            hyp = gt[keep] + rng.normal(0, jitter, (int(keep.sum()), 4))
            hyp[:, 2:] = np.maximum(hyp[:, 2:], 1.0)
            hid = gt_ids[keep].copy()
            for k in range(len(hid)):                          # persistent identity swaps
                if rng.random() < swap:
                    perm[hid[k]] = 1000 + int(rng.integers(0, 50))
                hid[k] = perm.get(hid[k], hid[k])
            _, first = np.unique(hid, return_index=True)       # hypotheses ids are unique within a frame
            hyp, hid = hyp[np.sort(first)], hid[np.sort(first)]
            nsp = rng.poisson(spurious)
            if nsp:
                sp = np.column_stack([rng.uniform(0, 1800, nsp), rng.uniform(0, 1000, nsp), rng.uniform(30, 120, nsp), rng.uniform(60, 250, nsp)])
                hyp = np.concatenate([hyp, sp]); hid = np.concatenate([hid, 5000 + rng.integers(0, 20, nsp)])
                _, first = np.unique(hid, return_index=True)
                hyp, hid = hyp[np.sort(first)], hid[np.sort(first)]
            d = mm.distances.iou_matrix(gt, hyp, max_iou=0.5)
            acc.update(gt_ids, hid, d)
            frames.append((gt_ids, gt, hid, hyp, d))
        seqs.append(frames); accs.append(acc)
    mh = mm.metrics.create()
    summ = mh.compute_many(accs, metrics=[m for m in CLEARMOT_METRICS], names=[f"s{i}" for i in range(len(accs))], generate_overall=True)
    blobs = {"metric_names": np.array(CLEARMOT_METRICS), "summary": summ[CLEARMOT_METRICS].to_numpy(dtype=np.float64),
             "row_names": np.array(list(summ.index))}
    for s, frames in enumerate(seqs):
        blobs[f"s{s}_offsets_gt"] = np.cumsum([0] + [len(f[0]) for f in frames]).astype(np.int64)
        blobs[f"s{s}_offsets_hyp"] = np.cumsum([0] + [len(f[2]) for f in frames]).astype(np.int64)
        blobs[f"s{s}_gt_ids"] = np.concatenate([f[0] for f in frames]); blobs[f"s{s}_gt_ltwh"] = np.concatenate([f[1] for f in frames])
        blobs[f"s{s}_hyp_ids"] = np.concatenate([f[2] for f in frames]); blobs[f"s{s}_hyp_ltwh"] = np.concatenate([f[3] for f in frames])
        blobs[f"s{s}_dist_f10"] = frames[10][4]                # one distance matrix per sequence for iou_distance_matrix
    np.savez_compressed(os.path.join(out_dir, "clearmot.npz"), **blobs)
    print(summ[["mota", "motp", "idf1", "num_switches", "num_fragmentations", "mostly_tracked"]])


def gen_mot_io(out_dir):
    """MOTChallenge text export: TrackingDataset.save_for_eval (tracklab/datastruct/tracking_dataset.py:161-236) loaded from its file
    with tracklab.utils (wandb / omegaconf imports) stubbed; inputs and the files it wrote are the fixture."""
    import importlib.util
    import tempfile
    import types
    for name in ("tracklab", "tracklab.utils"):
        if name not in sys.modules:
            m = types.ModuleType(name); m.__path__ = []; sys.modules[name] = m
    sys.modules["tracklab.utils"].wandb = types.SimpleNamespace(log=lambda *a, **k: None)
    spec = importlib.util.spec_from_file_location("ref_tracking_dataset", os.path.join(REF, "tracklab", "datastruct", "tracking_dataset.py"))
    td = importlib.util.module_from_spec(spec); spec.loader.exec_module(td)
    ds = object.__new__(td.TrackingDataset)
    rng = np.random.default_rng(4)
    nv, per = 3, [25, 1, 12]
    video = pd.DataFrame({"name": ["seq-A", "seq_B", "c"] + ["empty"]}, index=pd.Index([3, 7, 11, 20], name="id"))
    img_rows, det_rows = [], []
    iid, did = 1000, 50000
    for v, (vid, nfr) in enumerate(zip([3, 7, 11], per)):
        for f in range(nfr):
            img_rows.append((iid, f, vid))
            for k in range(int(rng.integers(0, 5))):
                box = rng.uniform(0, 900, 4).astype(np.float32) if v != 2 else rng.uniform(0, 900, 4)      # float32 and float64 boxes
                det_rows.append((did, iid, vid, box, float(np.round(rng.uniform(0.1, 1.0), int(rng.integers(1, 9)))),
                                 float(rng.integers(1, 9)) if rng.random() > 0.15 else np.nan, int(rng.integers(1, 4))))
                did += 1
            iid += 1
    imgs = pd.DataFrame({"frame": [r[1] for r in img_rows], "video_id": [r[2] for r in img_rows]}, index=pd.Index([r[0] for r in img_rows], name="id"))
    det = pd.DataFrame({"image_id": [r[1] for r in det_rows], "video_id": [r[2] for r in det_rows], "bbox_ltwh": [r[3] for r in det_rows],
                        "bbox_conf": [r[4] for r in det_rows], "track_id": [r[5] for r in det_rows], "category_id": [r[6] for r in det_rows]},
                       index=pd.Index([r[0] for r in det_rows], name="id"))
    det = det.sample(frac=1.0, random_state=1)                 # detections arrive in any order
    blobs = {}
    for tag, kw in (("plain", {}), ("classes", dict(save_classes=True))):
        with tempfile.TemporaryDirectory() as tmp:
            ds.save_for_eval(det, imgs.copy(), video, tmp, **kw)
            for name in video["name"]:
                blobs[f"{tag}_{name}"] = np.array(open(os.path.join(tmp, f"{name}.txt")).read())
    np.savez_compressed(os.path.join(out_dir, "mot_io.npz"), det_index=det.index.to_numpy(), det_image_id=det.image_id.to_numpy(),
                        det_video_id=det.video_id.to_numpy(), det_ltwh=np.stack([np.asarray(b, np.float64) for b in det.bbox_ltwh]),
                        det_is_f32=np.array([b.dtype == np.float32 for b in det.bbox_ltwh]), det_conf=det.bbox_conf.to_numpy(),
                        det_track_id=det.track_id.to_numpy(), det_category=det.category_id.to_numpy(), img_index=imgs.index.to_numpy(),
                        img_frame=imgs.frame.to_numpy(), img_video_id=imgs.video_id.to_numpy(), video_index=video.index.to_numpy(),
                        video_name=np.array(list(video["name"])), **blobs)
    print("mot_io:", {k: len(str(v)) for k, v in blobs.items()})


def gen_ssort_camera(out_dir):
    """Plain StrongSORT with `ecc: true` minus the estimator: Tracker.camera_update (sort/tracker.py:66-68) -> Track.camera_update
    (sort/track.py:221-239) run as is before every StrongSORT.update, like strong_sort_api.py:62-65, with Track.ECC (the
    cv2.findTransformECC call) replaced by a per-frame synthetic float32 (2,3) warp; one frame gets a far-off warp (get_matrix -> identity)."""
    ss, Metric, Tracker = _import_plain_strong_sort()
    import strong_sort.sort.track as track_mod
    hp = dict(max_dist=0.2, max_iou_dist=0.7, max_age=30, max_unmatched_preds=7, n_init=2, nn_budget=20, mc_lambda=0.995, ema_alpha=0.9)
    D, nframes = 32, 60
    rng = np.random.default_rng(11)
    warps = []
    for f in range(nframes):
        th = rng.normal(0, 0.004)
        wm = np.array([[np.cos(th), -np.sin(th), rng.normal(0, 3.0)], [np.sin(th), np.cos(th), rng.normal(0, 2.0)]], dtype=np.float32)
        if f == 25:
            wm[0, 2] = 500.0                                   # ||I - M|| >= 100: get_matrix falls back to the identity
        warps.append(wm)
    cur = {}
    orig = track_mod.Track.ECC
    track_mod.Track.ECC = lambda self, src, dst, *a, **k: (cur["w"].copy(), None)
    try:
        model = object.__new__(ss.StrongSORT)
        model.max_dist = hp["max_dist"]
        model.tracker = Tracker(Metric("cosine", hp["max_dist"], hp["nn_budget"]), max_iou_dist=hp["max_iou_dist"], max_age=hp["max_age"],
                                n_init=hp["n_init"], max_unmatched_preds=hp["max_unmatched_preds"], mc_lambda=hp["mc_lambda"], ema_alpha=hp["ema_alpha"])
        img = np.zeros((1080, 1920, 3), dtype=np.uint8)
        in_off, out_off, dets_all, embs, rows, blobs = [0], [0], [], [], [], {}
        prev = None
        for fr in SyntheticStream(21, 25, nframes, parts=1, dim=D, with_embeddings=True, miss_prob=0.15, churn_period=20):
            f = fr["frame"]
            dets, emb = fr["dets"], fr["embeddings"][:, 0, :].astype(np.float32)
            cur["w"] = warps[f]
            if prev is not None:                               # strong_sort_api.py:62-65
                model.tracker.camera_update(prev, img)
            prev = img
            dets_all.append(dets); embs.append(emb); in_off.append(in_off[-1] + len(dets))
            feats = torch.from_numpy(emb.copy())
            model._get_features = lambda xywhs, im, feats=feats: feats
            out = model.update(torch.from_numpy(dets.copy()), img)
            for r in out:
                rows.append([float(r[k]) for k in (0, 1, 2, 3, 4, 5, 6, 8)])
            out_off.append(out_off[-1] + len(out))
            if f in (1, 2, 10, 25, 26, 59):
                tr = model.tracker.tracks
                blobs[f"f{f}_track_ids"] = np.array([t.track_id for t in tr], dtype=np.int64)
                blobs[f"f{f}_mean"] = np.stack([np.asarray(t.mean, dtype=np.float64) for t in tr]).reshape(-1, 8)
    finally:
        track_mod.Track.ECC = orig
    np.savez_compressed(os.path.join(out_dir, "camera_ssort.npz"), dets=np.concatenate(dets_all), embeddings=np.concatenate(embs),
                        det_offsets=np.array(in_off, dtype=np.int64), out_offsets=np.array(out_off, dtype=np.int64),
                        rows=np.array(rows, dtype=np.float64).reshape(-1, 8), warps=np.stack(warps), config=json.dumps(hp), dim=D, **blobs)
    print(f"ssort_camera: rows_out={out_off[-1]} tracks={len(model.tracker.tracks)}")


def gen_botsort_gmc(out_dir):
    """BoT-SORT with camera-motion compensation minus the estimator: BoTSORT.update run as is with GMC.apply (the cv2 estimators,
    gmc.py) replaced by a synthetic (2,3) float64 warp per frame (small rotation + scale + translation), so that STrack.multi_gmc
    (bot_sort.py:93-109) runs on non-identity warps."""
    _import_byte_track()
    _import_plain_strong_sort()
    import bot_sort.bot_sort as bs
    from bot_sort.basetrack import BaseTrack
    from bot_sort.kalman_filter import KalmanFilter
    hp, D, nframes = dict(BOT_DEFAULTS, track_buffer=15), 32, 80
    model = object.__new__(bs.BoTSORT)
    model.tracked_stracks, model.lost_stracks, model.removed_stracks = [], [], []
    BaseTrack.clear_count()
    model.frame_id, model.lambda_, model.track_high_thresh, model.new_track_thresh = 0, hp["lambda_"], hp["track_high_thresh"], hp["new_track_thresh"]
    model.buffer_size = model.max_time_lost = int(hp["frame_rate"] / 30.0 * hp["track_buffer"])
    model.kalman_filter = KalmanFilter()
    model.proximity_thresh, model.appearance_thresh, model.match_thresh = hp["proximity_thresh"], hp["appearance_thresh"], hp["match_thresh"]
    rng = np.random.default_rng(31)
    warps = []
    for f in range(nframes):
        th, sc = rng.normal(0, 0.003), 1 + rng.normal(0, 0.002)
        warps.append(np.array([[sc * np.cos(th), -sc * np.sin(th), rng.normal(0, 2.5)], [sc * np.sin(th), sc * np.cos(th), rng.normal(0, 1.5)]]))
    cur = {}
    model.gmc = types.SimpleNamespace(apply=lambda img, dets: cur["w"].copy())
    img = np.zeros((1080, 1920, 3), dtype=np.uint8)
    in_off, out_off, dets_all, embs, rows, blobs = [0], [0], [], [], [], {}
    for fr in SyntheticStream(41, 30, nframes, parts=1, dim=D, with_embeddings=True, miss_prob=0.1, low_conf_frac=0.25, churn_period=25):
        f = fr["frame"]
        keep = fr["dets"][:, 4] > 0.4
        d, e = fr["dets"][keep], fr["embeddings"][keep, 0, :].astype(np.float32)
        dets_all.append(d); embs.append(e); in_off.append(in_off[-1] + len(d))
        cur["w"] = warps[f]
        hi = d[:, 4] > hp["track_high_thresh"]
        feats = torch.from_numpy(e[hi].copy())
        model._get_features = lambda xywh, im, feats=feats: feats
        out = model.update(torch.from_numpy(d.copy()), img)
        for r in out:
            rows.append([float(v) for v in r])
        out_off.append(out_off[-1] + len(out))
        if f in (1, 2, 20, 79):
            for lname, lst in (("trk", model.tracked_stracks), ("lost", model.lost_stracks)):
                blobs[f"f{f}_{lname}_ids"] = np.array([t.track_id for t in lst], dtype=np.int64)
                blobs[f"f{f}_{lname}_mean"] = np.array([np.asarray(t.mean, dtype=np.float64) for t in lst]).reshape(-1, 8)
                blobs[f"f{f}_{lname}_cov"] = np.array([np.asarray(t.covariance, dtype=np.float64) for t in lst]).reshape(-1, 8, 8)
    np.savez_compressed(os.path.join(out_dir, "gmc_botsort.npz"), dets=np.concatenate(dets_all), embeddings=np.concatenate(embs),
                        det_offsets=np.array(in_off, dtype=np.int64), out_offsets=np.array(out_off, dtype=np.int64),
                        rows=np.array(rows, dtype=np.float64).reshape(-1, 8), warps=np.stack(warps), config=json.dumps(hp), dim=D, **blobs)
    print(f"gmc_botsort: rows_out={out_off[-1]} next_id={BaseTrack._count + 1}")


def gen_deepocsort_cmc(out_dir):
    """Deep-OC-SORT with camera-motion compensation minus the estimator: OCSort.update run as is with cmc_off False and
    CMCComputer.compute_affine (cv2 optical flow, cmc.py) replaced by a synthetic (2,3) float64 warp per frame, so that
    KalmanBoxTracker.apply_affine_correction / KalmanFilterNew.apply_affine_correction (ocsort.py:261-281, kalmanfilter.py:387-405) run
    on non-identity warps -- including the double warp of a last observation that is still inside the delta_t window (one numpy array
    is stored as last_observation AND observations[age]) and the frozen state of unobserved tracks."""
    _install_filterpy_shim()
    _import_plain_strong_sort()
    saved_lap = sys.modules.get("lap", "absent")
    sys.modules["lap"] = None
    import deep_oc_sort.ocsort as doc
    try:
        hp = dict(DOC_DEFAULTS, det_thresh=0.3, delta_t=3, max_age=12, min_hits=1, cmc_off=False)
        D, nframes = 32, 90
        model = object.__new__(doc.OCSort)
        model.max_age, model.min_hits, model.iou_threshold = hp["max_age"], hp["min_hits"], hp["iou_threshold"]
        model.trackers, model.frame_count, model.det_thresh, model.delta_t = [], 0, hp["det_thresh"], hp["delta_t"]
        model.asso_func, model.inertia = doc.ASSO_FUNCS[hp["asso_func"]], hp["inertia"]
        model.w_association_emb, model.alpha_fixed_emb, model.aw_param = hp["w_association_emb"], hp["alpha_fixed_emb"], hp["aw_param"]
        doc.KalmanBoxTracker.count = 0
        model.embedding_off, model.cmc_off, model.aw_off, model.new_kf_off = False, False, False, False
        rng = np.random.default_rng(51)
        warps = []
        for f in range(nframes):
            th, sc = rng.normal(0, 0.003), 1 + rng.normal(0, 0.002)
            warps.append(np.array([[sc * np.cos(th), -sc * np.sin(th), rng.normal(0, 2.5)], [sc * np.sin(th), sc * np.cos(th), rng.normal(0, 1.5)]]))
        cur = {}
        model.cmc = types.SimpleNamespace(compute_affine=lambda img, dets, tag: cur["w"].copy())
        img = np.zeros((1080, 1920, 3), dtype=np.uint8)
        in_off, out_off, dets_all, embs, rows, blobs = [0], [0], [], [], [], {}
        for fr in SyntheticStream(61, 25, nframes, parts=1, dim=D, with_embeddings=True, miss_prob=0.2, churn_period=25):
            f = fr["frame"]
            keep = fr["dets"][:, 4] > 0.4
            d, e = fr["dets"][keep], fr["embeddings"][keep, 0, :].astype(np.float32)
            e = e / np.linalg.norm(e, axis=1, keepdims=True)
            dets_all.append(d); embs.append(e); in_off.append(in_off[-1] + len(d))
            cur["w"] = warps[f]
            thr = d[:, 4] > hp["det_thresh"]
            feats = torch.from_numpy(e[thr].copy())
            model._get_features = lambda xyxy, im, feats=feats: feats
            out = np.asarray(model.update(torch.from_numpy(d.copy()), img), dtype=np.float64).reshape(-1, 8)
            rows.extend(out.tolist())
            out_off.append(out_off[-1] + len(out))
            if f in (1, 2, 3, 10, 40, 89):
                T = model.trackers
                blobs[f"f{f}_ids"] = np.array([t.id for t in T], dtype=np.int64)
                blobs[f"f{f}_x"] = np.array([t.kf.x[:, 0] for t in T], dtype=np.float64).reshape(-1, 8)
                blobs[f"f{f}_P"] = np.array([t.kf.P for t in T], dtype=np.float64).reshape(-1, 8, 8)
                blobs[f"f{f}_last"] = np.array([t.last_observation for t in T], dtype=np.float64).reshape(-1, 5)
                blobs[f"f{f}_vel"] = np.array([t.velocity if t.velocity is not None else (0, 0) for t in T], dtype=np.float64).reshape(-1, 2)
                blobs[f"f{f}_state"] = np.array([[t.time_since_update, t.hits, t.hit_streak, t.age, int(t.frozen), int(t.kf.observed)]
                                                 for t in T], dtype=np.int64).reshape(-1, 6)
        np.savez_compressed(os.path.join(out_dir, "cmc_deepocsort.npz"), dets=np.concatenate(dets_all), embeddings=np.concatenate(embs),
                            det_offsets=np.array(in_off, dtype=np.int64), out_offsets=np.array(out_off, dtype=np.int64),
                            rows=np.array(rows, dtype=np.float64).reshape(-1, 8), warps=np.stack(warps),
                            config=json.dumps(dict(hp, cmc_off=True)), dim=D, **blobs)
        print(f"cmc_deepocsort: rows_out={out_off[-1]} next_id={doc.KalmanBoxTracker.count}")
    finally:
        if saved_lap == "absent":
            sys.modules.pop("lap", None)
        else:
            sys.modules["lap"] = saved_lap


def gen_ssort_setorder(out_dir):
    """A plain-StrongSORT run (found by tests/golden/fuzz_reference.py, trial 57) in which the iteration order of the CPython set behind
    `unmatched_tracks = list(set(track_indices) - set(k for k, _ in matches))` (sort/linear_assignment.py:126) is NOT ascending and decides
    the ids of two tracks born in the same frame: 32 objects, so track indices exceed the 8- / 32-entry tables of small result sets."""
    ss, Metric, Tracker = _import_plain_strong_sort()
    hp = dict(max_dist=0.24058051284092766, max_iou_dist=0.8946939275194115, max_age=10, max_unmatched_preds=2, n_init=3, nn_budget=11,
              mc_lambda=0.9660812618861851, ema_alpha=0.8694303929964391)
    D = 16
    model = object.__new__(ss.StrongSORT)
    model.max_dist = hp["max_dist"]
    model.tracker = Tracker(Metric("cosine", hp["max_dist"], hp["nn_budget"]), max_iou_dist=hp["max_iou_dist"], max_age=hp["max_age"],
                            n_init=hp["n_init"], max_unmatched_preds=hp["max_unmatched_preds"], mc_lambda=hp["mc_lambda"], ema_alpha=hp["ema_alpha"])
    img = np.zeros((1080, 1920, 3), dtype=np.uint8)
    in_off, out_off, dets_all, embs, rows = [0], [0], [], [], []
    for fr in SyntheticStream(4057, 32, 30, parts=1, dim=D, with_embeddings=True, miss_prob=0.05, low_conf_frac=0.2, churn_period=1000):
        d, e = fr["dets"], fr["embeddings"][:, 0, :].astype(np.float32)
        dets_all.append(d); embs.append(e); in_off.append(in_off[-1] + len(d))
        feats = torch.from_numpy(e.copy())
        model._get_features = lambda xywhs, im, feats=feats: feats
        out = model.update(torch.from_numpy(d.copy()), img)
        rows.extend([[float(r[k]) for k in (0, 1, 2, 3, 4, 5, 6, 8)] for r in out])
        out_off.append(out_off[-1] + len(out))
    np.savez_compressed(os.path.join(out_dir, "setorder_ssort.npz"), dets=np.concatenate(dets_all), embeddings=np.concatenate(embs),
                        det_offsets=np.array(in_off, dtype=np.int64), out_offsets=np.array(out_off, dtype=np.int64),
                        rows=np.array(rows, dtype=np.float64).reshape(-1, 8), config=json.dumps(hp), dim=D,
                        python=np.array(sys.version.split()[0]))
    print(f"setorder_ssort: rows_out={out_off[-1]} python {sys.version.split()[0]}")


def gen_nms(out_dir):
    """G2: non_max_suppression of strong_sort/sort/preprocessing.py:6-73 -- dead code in the reference (never called; `np.float` was removed from
    NumPy 1.24), run here with np.float restored. Distinct keys (np.argsort's default sort is not stable)."""
    _install_cv2_stub()
    had = hasattr(np, "float")
    if not had:
        np.float = float
    try:
        import strong_sort.sort.preprocessing as pre
        rng = np.random.default_rng(61)
        blobs = {}
        for ci, (n, thr, with_scores) in enumerate([(60, 0.5, True), (60, 0.5, False), (200, 0.3, True), (1, 0.5, True), (300, 0.8, False), (128, 1.0, True)]):
            c = rng.uniform(100, 1500, (max(n // 3, 1), 2))
            base = np.concatenate([c, rng.uniform(40, 120, (len(c), 1)), rng.uniform(90, 300, (len(c), 1))], 1)
            boxes = np.concatenate([base[rng.integers(0, len(base), n)][:, :2] + rng.normal(0, 12, (n, 2)), base[rng.integers(0, len(base), n)][:, 2:] * rng.uniform(0.8, 1.2, (n, 2))], 1)
            scores = rng.permutation(n).astype(np.float64) / n + 0.001 if with_scores else None
            pick = pre.non_max_suppression(boxes.copy(), thr, scores)
            blobs[f"c{ci}_boxes"], blobs[f"c{ci}_thr"], blobs[f"c{ci}_pick"] = boxes, np.array(thr), np.asarray(pick, dtype=np.int64)
            if with_scores:
                blobs[f"c{ci}_scores"] = scores
        blobs["n_cases"] = np.array(6)
        assert pre.non_max_suppression(np.zeros((0, 4)), 0.5) == []
    finally:
        if not had:
            del np.float
    np.savez_compressed(os.path.join(out_dir, "deepsort_nms.npz"), **blobs)
    print("deepsort_nms.npz", {k: v.shape for k, v in blobs.items() if k.endswith("pick")})


def main():
    out_dir = HERE
    only = set(sys.argv[1:])
    gens = {"ocsort": gen_ocsort, "iou": gen_iou_family, "kf7": gen_kf7, "lsa": gen_lsa,
            "coords": gen_coords, "bpbss": gen_bpbss, "kf8": gen_kf8, "hota": gen_hota, "cosine": gen_cosine, "motion": gen_motion_costs, "ssort": gen_ssort, "pil": gen_pil_preprocess, "pil_sweep": gen_pil_sweep, "bytetrack": gen_bytetrack, "botsort": gen_botsort, "deepocsort": gen_deepocsort, "clearmot": gen_clearmot, "mot_io": gen_mot_io, "ssort_camera": gen_ssort_camera, "botsort_gmc": gen_botsort_gmc, "deepocsort_cmc": gen_deepocsort_cmc, "ssort_setorder": gen_ssort_setorder, "nms": gen_nms}
    for k, fn in gens.items():
        if not only or k in only:
            fn(out_dir)
    with open(os.path.join(out_dir, "MANIFEST.json"), "w") as f:
        json.dump({"numpy": np.__version__, "scipy": scipy.__version__, "torch": torch.__version__,
                   "lsa_solver": "scipy.optimize.linear_sum_assignment (lap absent -> reference fallback)",
                   "reference": "TrackingLaboratory/tracklab v1.3.24 @ /root/reference",
                   "unpinned": ["torchreid.metrics.distance.compute_distance_matrix_using_bp_features "
                                "(restated in make_golden.py:bp_distance_restated)"]}, f, indent=1)


if __name__ == "__main__":
    main()
