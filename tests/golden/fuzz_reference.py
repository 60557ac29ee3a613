"""Differential fuzzing of the C oracle against the reference itself (THIS container only: imports /root/reference through the
same shims as make_golden.py). Not part of the test suite -- the committed goldens are; this widens the net over random
hyper-parameters and streams and prints the first divergence of each tracker, if any.
usage: python tests/golden/fuzz_reference.py [n_trials] [trackers: ocsort,bpbss,bytetrack,botsort,deepocsort,ssort,ssort_cam,botsort_gmc,deepocsort_cmc]"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as mg  # noqa: E402  (sets sys.path for the reference plugins and the repo)
import oracle  # noqa: E402
from tracklab_amd.synth import SyntheticStream, ltrb_to_ltwh_rows, synth_keypoints  # noqa: E402

oracle.build()
oracle.python_set_order(os.environ.get("ORC_PYTHON_SET_ORDER", "1") == "1")     # compare with the reference as CPython 3.10 runs it
N = int(sys.argv[1]) if len(sys.argv) > 1 else 10
WHICH = set(sys.argv[2].split(",")) if len(sys.argv) > 2 else {"ocsort", "bpbss", "bytetrack", "botsort", "deepocsort", "ssort"}
IMG = np.zeros((1080, 1920, 3), np.uint8)
BIG = int(os.environ.get("FUZZ_MAX_OBJECTS", "0"))          # > 0: streams with up to this many objects (crowded scenes)
FIRST = int(os.environ.get("FUZZ_FIRST", "0"))              # first trial index (trials are seeded by their index: ranges can run in parallel)
EXACT = {"rows": 0, "kf_bits_equal": 0}                     # r03: how many Kalman boxes are BIT-identical to the reference's (bpbss)


def nobj(rng, lo, hi):
    return int(rng.integers(lo, BIG if BIG > 0 else hi))


def stream_kw(rng):
    return dict(miss_prob=float(rng.choice([0.0, 0.05, 0.2])), low_conf_frac=float(rng.choice([0.0, 0.2, 0.4])),
                churn_period=int(rng.choice([20, 40, 1000])))


BOX_BITS = {}                                               # r03: per tracker, (output rows, rows whose four box coordinates are BIT-identical to the reference's)


def compare(name, trial, f, got, exp, tol=1e-9):
    exp = np.asarray(exp, dtype=np.float64).reshape(-1, 8)
    if got.shape == exp.shape and len(exp):
        c = BOX_BITS.setdefault(name, [0, 0])
        c[0] += len(exp); c[1] += int((got[:, :4] == exp[:, :4]).all(axis=1).sum())
    if got.shape != exp.shape or not np.array_equal(got[:, 4:], exp[:, 4:]) or not np.allclose(got[:, :4], exp[:, :4], rtol=tol, atol=tol):
        print(f"DIVERGENCE {name} trial {trial} frame {f}: shapes {got.shape} {exp.shape}")
        return False
    return True


def fuzz_bytetrack(trial, rng):
    mg._import_byte_track()
    from byte_track.basetrack import BaseTrack
    from byte_track.byte_tracker import BYTETracker
    hp = dict(track_thresh=float(rng.uniform(0.3, 0.7)), match_thresh=float(rng.uniform(0.5, 0.95)), track_buffer=int(rng.integers(3, 40)),
              frame_rate=int(rng.choice([15, 30])))
    BaseTrack._count = 0
    ref, orc = BYTETracker(**hp), oracle.ByteTrack(**hp)
    for fr in SyntheticStream(1000 + trial, nobj(rng, 5, 60), 120, **stream_kw(rng)):
        d = fr["dets"][fr["dets"][:, 4] > 0.4]
        if len(d) == 0:
            continue
        if not compare("bytetrack", trial, fr["frame"], orc.update(d), ref.update(torch.from_numpy(d.copy()), IMG)):
            return False
    return True


def fuzz_botsort(trial, rng):
    mg._import_byte_track(); mg._import_plain_strong_sort()
    import bot_sort.bot_sort as bs
    from bot_sort.basetrack import BaseTrack
    from bot_sort.gmc import GMC
    from bot_sort.kalman_filter import KalmanFilter
    D = int(rng.choice([16, 64]))
    hp = dict(track_high_thresh=float(rng.uniform(0.3, 0.7)), new_track_thresh=float(rng.uniform(0.3, 0.8)), track_buffer=int(rng.integers(3, 40)),
              match_thresh=float(rng.uniform(0.3, 0.9)), proximity_thresh=float(rng.uniform(0.3, 0.7)), appearance_thresh=float(rng.uniform(0.1, 0.5)),
              frame_rate=30, lambda_=float(rng.uniform(0.9, 0.995)))
    m = object.__new__(bs.BoTSORT)
    m.tracked_stracks, m.lost_stracks, m.removed_stracks = [], [], []
    BaseTrack.clear_count()
    m.frame_id, m.lambda_, m.track_high_thresh, m.new_track_thresh = 0, hp["lambda_"], hp["track_high_thresh"], hp["new_track_thresh"]
    m.buffer_size = m.max_time_lost = int(hp["frame_rate"] / 30.0 * hp["track_buffer"])
    m.kalman_filter = KalmanFilter()
    m.proximity_thresh, m.appearance_thresh, m.match_thresh = hp["proximity_thresh"], hp["appearance_thresh"], hp["match_thresh"]
    m.gmc = GMC(method="none", verbose=[None, False])
    orc = oracle.BoTSORT(D, **hp)
    for fr in SyntheticStream(2000 + trial, nobj(rng, 5, 50), 120, parts=1, dim=D, with_embeddings=True, **stream_kw(rng)):
        keep = fr["dets"][:, 4] > 0.4
        d, e = fr["dets"][keep], fr["embeddings"][keep, 0, :].astype(np.float32)
        if len(d) == 0:
            continue
        hi = d[:, 4] > hp["track_high_thresh"]
        feats = torch.from_numpy(e[hi].copy())
        m._get_features = lambda xywh, img, feats=feats: feats
        if not compare("botsort", trial, fr["frame"], orc.update(d, e), m.update(torch.from_numpy(d.copy()), IMG)):
            return False
    return True


def fuzz_deepocsort(trial, rng):
    mg._install_filterpy_shim(); mg._import_plain_strong_sort()
    saved = sys.modules.get("lap", "absent")
    sys.modules["lap"] = None
    try:
        import deep_oc_sort.ocsort as doc
        D = int(rng.choice([16, 64]))
        hp = dict(det_thresh=float(rng.choice([0.0, 0.3, 0.5])), max_age=int(rng.integers(3, 40)), min_hits=int(rng.integers(1, 4)),
                  iou_threshold=float(rng.uniform(0.15, 0.4)), delta_t=int(rng.integers(1, 4)), asso_func=str(rng.choice(["iou", "giou", "diou", "ciou"])),
                  inertia=float(rng.uniform(0.0, 0.5)), w_association_emb=float(rng.uniform(0.2, 1.0)), alpha_fixed_emb=float(rng.uniform(0.8, 0.98)),
                  aw_param=float(rng.uniform(0.3, 0.7)), embedding_off=False, cmc_off=True, aw_off=bool(rng.random() < 0.3), new_kf_off=False)
        m = object.__new__(doc.OCSort)
        m.max_age, m.min_hits, m.iou_threshold, m.trackers, m.frame_count = hp["max_age"], hp["min_hits"], hp["iou_threshold"], [], 0
        m.det_thresh, m.delta_t, m.asso_func, m.inertia = hp["det_thresh"], hp["delta_t"], doc.ASSO_FUNCS[hp["asso_func"]], hp["inertia"]
        m.w_association_emb, m.alpha_fixed_emb, m.aw_param = hp["w_association_emb"], hp["alpha_fixed_emb"], hp["aw_param"]
        doc.KalmanBoxTracker.count = 0
        m.embedding_off, m.cmc_off, m.aw_off, m.new_kf_off = False, True, hp["aw_off"], False
        orc = oracle.DeepOCSort(D, **hp)
        normed = rng.random() < 0.7
        for fr in SyntheticStream(3000 + trial, nobj(rng, 5, 50), 120, parts=1, dim=D, with_embeddings=True, **stream_kw(rng)):
            keep = fr["dets"][:, 4] > 0.4
            d, e = fr["dets"][keep], fr["embeddings"][keep, 0, :].astype(np.float32)
            if normed and len(e):
                e = e / np.linalg.norm(e, axis=1, keepdims=True)
            if len(d) == 0:
                continue
            thr = d[:, 4] > hp["det_thresh"]
            feats = torch.from_numpy(e[thr].copy())
            m._get_features = lambda xyxy, img, feats=feats: feats
            if not compare("deepocsort", trial, fr["frame"], orc.update(d, e), np.asarray(m.update(torch.from_numpy(d.copy()), IMG)), tol=0):
                return False
        return True
    finally:
        if saved == "absent":
            sys.modules.pop("lap", None)
        else:
            sys.modules["lap"] = saved


def fuzz_ssort(trial, rng):
    ss, Metric, Tracker = mg._import_plain_strong_sort()
    D = int(rng.choice([16, 64]))
    hp = dict(max_dist=float(rng.uniform(0.1, 0.4)), max_iou_dist=float(rng.uniform(0.5, 0.9)), max_age=int(rng.integers(3, 40)),
              max_unmatched_preds=int(rng.integers(0, 8)), n_init=int(rng.integers(1, 4)), nn_budget=int(rng.integers(2, 30)),
              mc_lambda=float(rng.uniform(0.9, 0.999)), ema_alpha=float(rng.uniform(0.8, 0.95)))
    m = object.__new__(ss.StrongSORT)
    m.max_dist = hp["max_dist"]
    m.tracker = Tracker(Metric("cosine", hp["max_dist"], hp["nn_budget"]), max_iou_dist=hp["max_iou_dist"], max_age=hp["max_age"],
                        n_init=hp["n_init"], max_unmatched_preds=hp["max_unmatched_preds"], mc_lambda=hp["mc_lambda"], ema_alpha=hp["ema_alpha"])
    orc = oracle.PlainStrongSORT(D, **hp, img_w=1920, img_h=1080)
    for fr in SyntheticStream(4000 + trial, nobj(rng, 5, 40), 100, parts=1, dim=D, with_embeddings=True, **stream_kw(rng)):
        d, e = fr["dets"], fr["embeddings"][:, 0, :].astype(np.float32)
        if len(d) == 0:
            continue
        m._get_features = lambda xywh, img, e=e: torch.from_numpy(e.copy())
        out = m.update(torch.from_numpy(d.copy()), IMG)                     # rows of 9: [x1, y1, x2, y2, id, cls, conf, queue, tracklab id]
        exp = np.array([[float(r[k]) for k in (0, 1, 2, 3, 4, 5, 6, 8)] for r in out], dtype=np.float64).reshape(-1, 8)
        if not compare("ssort", trial, fr["frame"], orc.update(d, e), exp, tol=0):
            return False
    return True


def fuzz_ocsort(trial, rng):
    mg._install_filterpy_shim()
    sys.modules["lap"] = None                                   # `import lap` fails -> scipy fallback, as in the reference's environment
    import oc_sort.ocsort as ref
    hp = dict(det_thresh=float(rng.choice([0.0, 0.3, 0.5])), max_age=int(rng.integers(3, 40)), min_hits=int(rng.integers(1, 4)),
              iou_threshold=float(rng.uniform(0.15, 0.4)), delta_t=int(rng.integers(1, 4)), asso_func=str(rng.choice(["iou", "giou", "diou", "ciou", "ct_dist"])),
              inertia=float(rng.uniform(0.0, 0.5)), use_byte=bool(rng.random() < 0.4))
    trk, orc = ref.OCSort(**hp), oracle.OCSort(**hp)
    for fr in SyntheticStream(5000 + trial, nobj(rng, 5, 60), 120, **stream_kw(rng)):
        d = fr["dets"]
        if len(d) == 0:
            continue
        inp = torch.from_numpy(d.copy())
        exp = np.asarray(trk.update(inp[inp[:, 4] > 0.4], None), dtype=np.float64).reshape(-1, 8)
        if not compare("ocsort", trial, fr["frame"], oracle.ocsort_wrapper_step(orc, d, 0.4), exp, tol=1e-9):
            return False
    return True


def fuzz_bpbss(trial, rng):
    mg._install_cv2_stub()
    import bpbreid_strong_sort.sort.nn_matching as nnm
    nnm.compute_distance_matrix_using_bp_features = mg.bp_distance_restated
    import bpbreid_strong_sort.strong_sort as ss
    K, D = int(rng.choice([3, 6])), int(rng.choice([16, 32]))
    oks = bool(rng.random() < 0.3)
    cfg = dict(mg.BPB_YAML, ema_alpha=float(rng.uniform(0.5, 0.95)), max_dist=float(rng.uniform(0.25, 0.6)), max_iou_distance=float(rng.uniform(0.6, 0.9)),
               max_age=int(rng.integers(5, 60)), n_init=int(rng.integers(0, 4)), min_bbox_confidence=float(rng.choice([0.0, 0.5])),
               only_position_for_kf_gating=bool(rng.random() < 0.3), max_kalman_prediction_without_update=int(rng.integers(0, 8)),
               matching_strategy=str(rng.choice(["strong_sort_matching", "bot_sort_matching"])), gating_thres_factor=float(rng.choice([1, 1.5])),
               motion_criterium="oks" if oks else "iou")
    model, orc = ss.StrongSORT(**cfg), oracle.StrongSORT(K, D, **cfg)
    kp_rng = np.random.default_rng(77 + trial)
    for fr in SyntheticStream(6000 + trial, nobj(rng, 5, 40), 100, parts=K, dim=D, with_embeddings=True, **stream_kw(rng)):
        d = fr["dets"]
        if len(d) == 0:
            continue
        ltwh, conf, did = ltrb_to_ltwh_rows(d[:, :4]), d[:, 4].copy(), d[:, 6].astype(np.int64)
        kp = synth_keypoints(kp_rng, d[:, :4]) if oks else None
        df = model.update(torch.from_numpy(did), torch.from_numpy(ltwh), torch.from_numpy(fr["embeddings"]), torch.from_numpy(fr["visibility"]),
                          torch.from_numpy(conf), torch.zeros(len(d), dtype=torch.float64), torch.ones(len(d), dtype=torch.float64) * fr["frame"],
                          torch.from_numpy(kp) if oks else None)
        got = orc.update(did, ltwh, fr["embeddings"], fr["visibility"], conf, keypoints=kp)
        exp_idx = np.array([int(i) for i in df.index], dtype=np.int64)
        exp_tid = np.array([int(t) for t in df.track_id], dtype=np.int64) if len(df) else np.zeros(0, np.int64)
        ok = len(got) == len(df) and np.array_equal(got["det_id"], exp_idx) and np.array_equal(got["track_id"], exp_tid)
        if ok and len(df):
            ref_kf = np.stack([np.asarray(b, dtype=np.float64) for b in df.track_bbox_kf_ltwh])
            EXACT["rows"] += len(df); EXACT["kf_bits_equal"] += int((got["kf_ltwh"] == ref_kf).all(axis=1).sum())
            ok = np.allclose(got["kf_ltwh"], ref_kf, rtol=1e-7, atol=1e-7) and \
                np.array_equal(got["hits"], df.hits.to_numpy().astype(np.int32)) and np.array_equal(got["tsu"], df.time_since_update.to_numpy().astype(np.int32))
        if not ok:
            print(f"DIVERGENCE bpbss trial {trial} frame {fr['frame']} cfg {cfg}")
            return False
    return True


def _rand_warp(rng, big=False):
    th, sc = rng.normal(0, 0.004), 1 + rng.normal(0, 0.003)
    return np.array([[sc * np.cos(th), -sc * np.sin(th), rng.normal(0, 200.0 if big else 3.0)], [sc * np.sin(th), sc * np.cos(th), rng.normal(0, 2.0)]])


def fuzz_ssort_cam(trial, rng):
    """plain StrongSORT with Tracker.camera_update before every update (strong_sort_api.py:62-65), Track.ECC patched to a random warp."""
    ss, Metric, Tracker = mg._import_plain_strong_sort()
    import strong_sort.sort.track as track_mod
    D = 16
    hp = dict(max_dist=float(rng.uniform(0.1, 0.4)), max_iou_dist=float(rng.uniform(0.5, 0.9)), max_age=int(rng.integers(3, 40)),
              max_unmatched_preds=int(rng.integers(0, 8)), n_init=int(rng.integers(1, 4)), nn_budget=int(rng.integers(2, 30)),
              mc_lambda=float(rng.uniform(0.9, 0.999)), ema_alpha=float(rng.uniform(0.8, 0.95)))
    cur = {}
    orig = track_mod.Track.ECC
    track_mod.Track.ECC = lambda self, src, dst, *a, **k: (cur["w"].copy(), None)
    try:
        m = object.__new__(ss.StrongSORT)
        m.max_dist = hp["max_dist"]
        m.tracker = Tracker(Metric("cosine", hp["max_dist"], hp["nn_budget"]), max_iou_dist=hp["max_iou_dist"], max_age=hp["max_age"],
                            n_init=hp["n_init"], max_unmatched_preds=hp["max_unmatched_preds"], mc_lambda=hp["mc_lambda"], ema_alpha=hp["ema_alpha"])
        orc = oracle.PlainStrongSORT(D, **hp, img_w=1920, img_h=1080)
        prev = None
        for fr in SyntheticStream(7000 + trial, nobj(rng, 5, 40), 80, parts=1, dim=D, with_embeddings=True, **stream_kw(rng)):
            d, e = fr["dets"], fr["embeddings"][:, 0, :].astype(np.float32)
            cur["w"] = _rand_warp(rng, big=rng.random() < 0.05).astype(np.float32)
            if prev is not None:
                m.tracker.camera_update(prev, IMG); orc.camera_update(cur["w"])
            prev = IMG
            if len(d) == 0:
                continue
            m._get_features = lambda xywh, img, e=e: torch.from_numpy(e.copy())
            out = m.update(torch.from_numpy(d.copy()), IMG)
            exp = np.array([[float(r[k]) for k in (0, 1, 2, 3, 4, 5, 6, 8)] for r in out], dtype=np.float64).reshape(-1, 8)
            if not compare("ssort_cam", trial, fr["frame"], orc.update(d, e), exp, tol=0):
                return False
        return True
    finally:
        track_mod.Track.ECC = orig


def fuzz_botsort_gmc(trial, rng):
    import types
    mg._import_byte_track(); mg._import_plain_strong_sort()
    import bot_sort.bot_sort as bs
    from bot_sort.basetrack import BaseTrack
    from bot_sort.kalman_filter import KalmanFilter
    D = 16
    hp = dict(track_high_thresh=float(rng.uniform(0.3, 0.7)), new_track_thresh=float(rng.uniform(0.3, 0.8)), track_buffer=int(rng.integers(3, 40)),
              match_thresh=float(rng.uniform(0.3, 0.9)), proximity_thresh=float(rng.uniform(0.3, 0.7)), appearance_thresh=float(rng.uniform(0.1, 0.5)),
              frame_rate=30, lambda_=float(rng.uniform(0.9, 0.995)))
    m = object.__new__(bs.BoTSORT)
    m.tracked_stracks, m.lost_stracks, m.removed_stracks = [], [], []
    BaseTrack.clear_count()
    m.frame_id, m.lambda_, m.track_high_thresh, m.new_track_thresh = 0, hp["lambda_"], hp["track_high_thresh"], hp["new_track_thresh"]
    m.buffer_size = m.max_time_lost = int(hp["frame_rate"] / 30.0 * hp["track_buffer"])
    m.kalman_filter = KalmanFilter()
    m.proximity_thresh, m.appearance_thresh, m.match_thresh = hp["proximity_thresh"], hp["appearance_thresh"], hp["match_thresh"]
    cur = {}
    m.gmc = types.SimpleNamespace(apply=lambda img, dets: cur["w"].copy())
    orc = oracle.BoTSORT(D, **hp)
    for fr in SyntheticStream(8000 + trial, nobj(rng, 5, 40), 80, parts=1, dim=D, with_embeddings=True, **stream_kw(rng)):
        keep = fr["dets"][:, 4] > 0.4
        d, e = fr["dets"][keep], fr["embeddings"][keep, 0, :].astype(np.float32)
        if len(d) == 0:
            continue
        cur["w"] = _rand_warp(rng)
        hi = d[:, 4] > hp["track_high_thresh"]
        feats = torch.from_numpy(e[hi].copy())
        m._get_features = lambda xywh, img, feats=feats: feats
        if not compare("botsort_gmc", trial, fr["frame"], orc.update(d, e, warp=cur["w"]), m.update(torch.from_numpy(d.copy()), IMG), tol=1e-9):
            return False
    return True


def fuzz_deepocsort_cmc(trial, rng):
    import types
    mg._install_filterpy_shim(); mg._import_plain_strong_sort()
    saved = sys.modules.get("lap", "absent")
    sys.modules["lap"] = None
    try:
        import deep_oc_sort.ocsort as doc
        D = 16
        hp = dict(det_thresh=float(rng.choice([0.0, 0.3, 0.5])), max_age=int(rng.integers(3, 30)), min_hits=1, iou_threshold=float(rng.uniform(0.15, 0.4)),
                  delta_t=int(rng.integers(1, 6)), asso_func=str(rng.choice(["iou", "giou", "diou", "ciou"])), inertia=float(rng.uniform(0.0, 0.5)),
                  w_association_emb=float(rng.uniform(0.2, 1.0)), alpha_fixed_emb=float(rng.uniform(0.8, 0.98)), aw_param=float(rng.uniform(0.3, 0.7)),
                  embedding_off=False, cmc_off=True, aw_off=bool(rng.random() < 0.3), new_kf_off=False)
        m = object.__new__(doc.OCSort)
        m.max_age, m.min_hits, m.iou_threshold, m.trackers, m.frame_count = hp["max_age"], hp["min_hits"], hp["iou_threshold"], [], 0
        m.det_thresh, m.delta_t, m.asso_func, m.inertia = hp["det_thresh"], hp["delta_t"], doc.ASSO_FUNCS[hp["asso_func"]], hp["inertia"]
        m.w_association_emb, m.alpha_fixed_emb, m.aw_param = hp["w_association_emb"], hp["alpha_fixed_emb"], hp["aw_param"]
        doc.KalmanBoxTracker.count = 0
        m.embedding_off, m.cmc_off, m.aw_off, m.new_kf_off = False, False, hp["aw_off"], False
        cur = {}
        m.cmc = types.SimpleNamespace(compute_affine=lambda img, dets, tag: cur["w"].copy())
        orc = oracle.DeepOCSort(D, **hp)
        for fr in SyntheticStream(9000 + trial, nobj(rng, 5, 40), 80, parts=1, dim=D, with_embeddings=True, **stream_kw(rng)):
            keep = fr["dets"][:, 4] > 0.4
            d, e = fr["dets"][keep], fr["embeddings"][keep, 0, :].astype(np.float32)
            if len(d) == 0:
                continue
            e = e / np.linalg.norm(e, axis=1, keepdims=True)
            cur["w"] = _rand_warp(rng)
            thr = d[:, 4] > hp["det_thresh"]
            feats = torch.from_numpy(e[thr].copy())
            m._get_features = lambda xyxy, img, feats=feats: feats
            if not compare("deepocsort_cmc", trial, fr["frame"], orc.update(d, e, warp=cur["w"]), np.asarray(m.update(torch.from_numpy(d.copy()), IMG)), tol=1e-9):
                return False
        return True
    finally:
        if saved == "absent":
            sys.modules.pop("lap", None)
        else:
            sys.modules["lap"] = saved


FUZZ = {"ssort_cam": fuzz_ssort_cam, "botsort_gmc": fuzz_botsort_gmc, "deepocsort_cmc": fuzz_deepocsort_cmc, "ocsort": fuzz_ocsort, "bpbss": fuzz_bpbss, "bytetrack": fuzz_bytetrack, "botsort": fuzz_botsort, "deepocsort": fuzz_deepocsort, "ssort": fuzz_ssort}
for name in sorted(WHICH):
    ok = 0
    for t in range(FIRST, FIRST + N):
        try:
            ok += bool(FUZZ[name](t, np.random.default_rng(9000 + t)))
        except Exception as ex:                             # a reference-side crash on odd hyper-parameters is reported, not fatal
            print(f"EXCEPTION {name} trial {t}: {type(ex).__name__}: {ex}")
    print(f"{name}: {ok}/{N} trials identical to the reference" + (f" (trials {FIRST}..{FIRST + N - 1})" if FIRST else "") +
          (f"; Kalman boxes bit-identical: {EXACT['kf_bits_equal']}/{EXACT['rows']} rows" if name == "bpbss" else "") +
          (f"; output boxes bit-identical: {BOX_BITS[name][1]}/{BOX_BITS[name][0]} rows" if name in BOX_BITS else ""))
