#!/usr/bin/env python3
"""Differential fuzz ON THE GPU BOX: every tracker bank of libtlk (through the C ABI) against the C oracle, frame by frame, on random hyper-parameter
sets x random streams (the generators of tests/golden/fuzz_reference.py, which fuzzes the oracle against the reference itself where /root/reference
exists -- together: HIP == oracle == reference). Crowded by default (up to FUZZ_MAX_OBJECTS objects, default 130). Rows must be EQUAL (ids, classes,
counters, boxes bit for bit); where the bank exposes it the Kalman state is compared bit for bit too. The oracle is the checker here, never the product.

    python tools/fuzz_gpu.py [trials per tracker] [tracker ...]      # ocsort bytetrack bpbss botsort deepocsort ssort
"""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import oracle                                                             # noqa: E402
from tracklab_amd import _lib                                             # noqa: E402
from tracklab_amd.synth import SyntheticStream, ltrb_to_ltwh_rows         # noqa: E402

oracle.build()
N = int(sys.argv[1]) if len(sys.argv) > 1 else 10
WHICH = sys.argv[2:] or ["ocsort", "bytetrack", "bpbss", "botsort", "deepocsort", "ssort"]
BIG = int(os.environ.get("FUZZ_MAX_OBJECTS", "130"))
# FUZZ_BIG_CAPACITY=1 (with FUZZ_MAX_OBJECTS of a few hundred): banks allocated at 4096 tracks / 512 detections, so that crowded trials leave the
# LDS tiers and run their per-frame lists and assignment problems out of HBM (r04 capacity tiers)
BIGCAP = os.environ.get("FUZZ_BIG_CAPACITY", "0") != "0"


def cap(t, d):
    return dict(max_tracks=4096, max_dets=512) if BIGCAP else dict(max_tracks=t, max_dets=d)
STATS = {}


def nobj(rng, lo):
    return int(rng.integers(lo, BIG))


def stream_kw(rng):
    return dict(miss_prob=float(rng.choice([0.0, 0.05, 0.2])), low_conf_frac=float(rng.choice([0.0, 0.2, 0.4])), churn_period=int(rng.choice([20, 40, 1000])))


def same(name, trial, f, what, a, b, exact=True, atol=0.0):
    a, b = np.asarray(a), np.asarray(b)
    ok = a.shape == b.shape and (np.array_equal(a, b, equal_nan=True) if exact else np.allclose(a, b, rtol=0, atol=atol, equal_nan=True))
    if not ok:
        print(f"DIVERGENCE {name} trial {trial} frame {f}: {what} shapes {a.shape} {b.shape}")
    return ok


def count(name, rows, state_rows=0):
    c = STATS.setdefault(name, {"rows": 0, "kalman_states": 0})
    c["rows"] += int(rows); c["kalman_states"] += int(state_rows)


def fuzz_ocsort(t, rng):
    hp = dict(det_thresh=float(rng.choice([0.0, 0.3, 0.5])), max_age=int(rng.integers(3, 40)), min_hits=int(rng.integers(1, 4)),
              iou_threshold=float(rng.uniform(0.15, 0.4)), delta_t=int(rng.integers(1, 4)), asso_func=str(rng.choice(["iou", "giou", "diou", "ciou", "ct_dist"])),
              inertia=float(rng.uniform(0.0, 0.5)), use_byte=bool(rng.random() < 0.4))
    bank, ref = _lib.OCSortBank(**hp, min_confidence=0.4, wrapper_mode=True, **cap(512, 256)), oracle.OCSort(**hp)
    try:
        for fr in SyntheticStream(5000 + t, nobj(rng, 5), 120, **stream_kw(rng)):
            d = fr["dets"]
            exp, got = oracle.ocsort_wrapper_step(ref, d, 0.4), bank.update(d, 0)
            if not same("ocsort", t, fr["frame"], "rows", got, exp):
                return False
            x, P, ids = bank.tracks(0)
            cx, cP, cids = ref.tracks()
            # asso_func ciou / ct_dist go through atan / a global maximum whose last bits may differ between libm and the device: the ROWS above must
            # still be equal; the Kalman state is compared bit for bit for the other three
            if hp["asso_func"] in ("iou", "giou", "diou") and not (same("ocsort", t, fr["frame"], "ids", ids, cids) and same("ocsort", t, fr["frame"], "x", x, cx) and same("ocsort", t, fr["frame"], "P", P, cP)):
                return False
            count("ocsort", len(exp), len(ids))
        return True
    finally:
        bank.close()


def fuzz_bytetrack(t, rng):
    hp = dict(track_thresh=float(rng.uniform(0.3, 0.7)), match_thresh=float(rng.uniform(0.5, 0.95)), track_buffer=int(rng.integers(3, 40)), frame_rate=int(rng.choice([15, 30])))
    bank, ref = _lib.ByteTrackBank(**hp, **cap(384, 128)), oracle.ByteTrack(**hp)
    try:
        for fr in SyntheticStream(1000 + t, min(nobj(rng, 5), 120), 120, **stream_kw(rng)):
            d = fr["dets"][fr["dets"][:, 4] > 0.4]
            if len(d) == 0:
                continue
            r = bank.update(d)
            got = np.column_stack([r["ltrb"], r["track_id"], r["cls"], r["score"], r["det_id"]]).astype(np.float64).reshape(-1, 8)
            if not same("bytetrack", t, fr["frame"], "rows", got, ref.update(d)):
                return False
            n_state = 0
            for which in (0, 1):
                gi, gm, gc, gs = bank.tracks(which)
                ci, cm, cc, cs = ref.tracks(which)
                if not (same("bytetrack", t, fr["frame"], "ids", gi, ci) and same("bytetrack", t, fr["frame"], "mean", gm, cm) and same("bytetrack", t, fr["frame"], "cov", gc, cc)):
                    return False
                n_state += len(gi)
            count("bytetrack", len(got), n_state)
        return True
    finally:
        bank.close()


def fuzz_bpbss(t, rng):
    K, D = int(rng.choice([3, 6])), int(rng.choice([16, 32, 256]))
    cfg = dict(ema_alpha=float(rng.uniform(0.5, 0.95)), mc_lambda=0.995, max_dist=float(rng.uniform(0.25, 0.6)), motion_criterium="iou",
               max_iou_distance=float(rng.uniform(0.6, 0.9)), max_oks_distance=0.7, max_age=int(rng.integers(5, 60)), n_init=int(rng.integers(0, 4)), nn_budget=100,
               min_bbox_confidence=float(rng.choice([0.0, 0.5])), only_position_for_kf_gating=bool(rng.random() < 0.3),
               max_kalman_prediction_without_update=int(rng.integers(0, 8)), matching_strategy=str(rng.choice(["strong_sort_matching", "bot_sort_matching"])),
               gating_thres_factor=float(rng.choice([1, 1.5])), w_kfgd=1, w_reid=1, w_st=1)
    bank, ref = _lib.BpbssBank(K, D, **cfg, **cap(1024, 256)), oracle.StrongSORT(K, D, **cfg)
    try:
        for fr in SyntheticStream(6000 + t, nobj(rng, 5), 100, parts=K, dim=D, with_embeddings=True, **stream_kw(rng)):
            d = fr["dets"]
            if len(d) == 0:
                continue
            ltwh, conf, ids = ltrb_to_ltwh_rows(d[:, :4]), d[:, 4].copy(), d[:, 6].astype(np.int64)
            exp = ref.update(ids, ltwh, fr["embeddings"], fr["visibility"], conf)
            got = bank.update(ids, ltwh, fr["embeddings"], fr["visibility"], conf)
            for name in ("det_id", "track_id", "hits", "age", "tsu", "state", "matched_name", "pred_valid", "kf_ltwh"):
                if not same("bpbss", t, fr["frame"], name, got[name], exp[name]):
                    return False
            if not same("bpbss", t, fr["frame"], "matched_dist", got["matched_dist"], exp["matched_dist"], exact=False, atol=1e-5):
                return False
            count("bpbss", len(exp), len(exp))
        return True
    finally:
        bank.close()


def fuzz_botsort(t, rng):
    D = int(rng.choice([16, 64]))
    hp = dict(track_high_thresh=float(rng.uniform(0.3, 0.7)), new_track_thresh=float(rng.uniform(0.3, 0.8)), track_buffer=int(rng.integers(3, 40)),
              match_thresh=float(rng.uniform(0.3, 0.9)), proximity_thresh=float(rng.uniform(0.3, 0.7)), appearance_thresh=float(rng.uniform(0.1, 0.5)),
              frame_rate=30, lambda_=float(rng.uniform(0.9, 0.995)))
    bank, ref = _lib.BoTSORTBank(D, **hp, **cap(384, 128)), oracle.BoTSORT(D, **hp)
    try:
        for fr in SyntheticStream(2000 + t, min(nobj(rng, 5), 120), 120, parts=1, dim=D, with_embeddings=True, **stream_kw(rng)):
            keep = fr["dets"][:, 4] > 0.4
            d, e = fr["dets"][keep], fr["embeddings"][keep, 0, :].astype(np.float32)
            if len(d) == 0:
                continue
            r = bank.update(d, e)
            got = np.column_stack([r["ltrb"], r["track_id"], r["cls"], r["score"], r["det_id"]]).astype(np.float64).reshape(-1, 8)
            if not same("botsort", t, fr["frame"], "rows", got, ref.update(d, e)):
                return False
            n_state = 0
            for which in (0, 1):
                gi, gm, gc, gs, gf = bank.tracks(which)
                ci, cm, cc, cs, cf = ref.tracks(which)
                if not (same("botsort", t, fr["frame"], "ids", gi, ci) and same("botsort", t, fr["frame"], "mean", gm, cm) and same("botsort", t, fr["frame"], "cov", gc, cc)):
                    return False
                n_state += len(gi)
            count("botsort", len(got), n_state)
        return True
    finally:
        bank.close()


def fuzz_deepocsort(t, rng):
    D = int(rng.choice([16, 64]))
    hp = dict(det_thresh=float(rng.choice([0.0, 0.3, 0.5])), max_age=int(rng.integers(3, 40)), min_hits=int(rng.integers(1, 4)),
              iou_threshold=float(rng.uniform(0.15, 0.4)), delta_t=int(rng.integers(1, 4)), asso_func=str(rng.choice(["iou", "giou", "diou", "ciou"])),
              inertia=float(rng.uniform(0.0, 0.5)), w_association_emb=float(rng.uniform(0.2, 1.0)), alpha_fixed_emb=float(rng.uniform(0.8, 0.98)),
              aw_param=float(rng.uniform(0.3, 0.7)), embedding_off=False, cmc_off=True, aw_off=bool(rng.random() < 0.3), new_kf_off=False)
    bank, ref = _lib.DeepOCSortBank(D, **hp, **cap(384, 128)), oracle.DeepOCSort(D, **hp)
    normed = rng.random() < 0.7
    try:
        for fr in SyntheticStream(3000 + t, min(nobj(rng, 5), 120), 120, parts=1, dim=D, with_embeddings=True, **stream_kw(rng)):
            keep = fr["dets"][:, 4] > 0.4
            d, e = fr["dets"][keep], fr["embeddings"][keep, 0, :].astype(np.float32)
            if normed and len(e):
                e = e / np.linalg.norm(e, axis=1, keepdims=True)
            if len(d) == 0:
                continue
            got, exp = bank.update(d, e), ref.update(d, e)
            if not same("deepocsort", t, fr["frame"], "rows", got, exp):
                return False
            gi, gx, gP, ge, gs, gv, gl = bank.tracks()
            ci, cx, cP, ce, cs, cv, cl = ref.tracks()
            if not (same("deepocsort", t, fr["frame"], "ids", gi, ci) and same("deepocsort", t, fr["frame"], "x", gx, cx) and same("deepocsort", t, fr["frame"], "P", gP, cP)):
                return False
            count("deepocsort", len(exp), len(gi))
        return True
    finally:
        bank.close()


def fuzz_ssort(t, rng):
    D = 64
    hp = dict(max_dist=float(rng.uniform(0.1, 0.4)), max_iou_dist=float(rng.uniform(0.5, 0.9)), max_age=int(rng.integers(3, 40)),
              max_unmatched_preds=int(rng.integers(0, 8)), n_init=int(rng.integers(1, 4)), nn_budget=int(rng.integers(2, 30)),
              mc_lambda=float(rng.uniform(0.9, 0.999)), ema_alpha=float(rng.uniform(0.8, 0.95)))
    bank, ref = _lib.SsortBank(D, **hp, **cap(512, 256)), oracle.PlainStrongSORT(D, **hp, img_w=1920, img_h=1080)
    try:
        for fr in SyntheticStream(4000 + t, min(nobj(rng, 5), 60), 60, parts=1, dim=D, with_embeddings=True, **stream_kw(rng)):      # (the C oracle of this tracker is the slow side)
            d, e = fr["dets"], fr["embeddings"][:, 0, :].astype(np.float32)
            if len(d) == 0:
                continue
            r = bank.update(d, e)
            got = np.column_stack([r["ltrb"], r["track_id"], r["class_id"], r["conf"], r["det_id"]]).astype(np.float64).reshape(-1, 8)
            if not same("ssort", t, fr["frame"], "rows", got, ref.update(d, e)):
                return False
            gi, gm, gc, gf, gs, gg = bank.tracks()
            ci, cm, cc, cf, cs, cg = ref.tracks()
            if not (same("ssort", t, fr["frame"], "ids", gi, ci) and same("ssort", t, fr["frame"], "mean", gm, cm) and same("ssort", t, fr["frame"], "cov", gc, cc)):
                return False
            count("ssort", len(got), len(gi))
        return True
    finally:
        bank.close()


def resnet_load():
    """FUZZ_LOAD=1: a bf16 ResNet-50 forward keeps the matrix cores busy from a side stream for the whole run -- the condition the product runs its
    association kernels in (tests/test_gpu_under_load.py; DESIGN.md section 2: the packed-FP32 finding only ever showed under such a load)."""
    import threading

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
                side.synchronize()
                forwards[0] += 4
    th = threading.Thread(target=run, daemon=True)
    th.start()
    return stop, th, forwards


LOAD = resnet_load() if os.environ.get("FUZZ_LOAD", "0") == "1" else None
FUZZ = {"ocsort": fuzz_ocsort, "bytetrack": fuzz_bytetrack, "bpbss": fuzz_bpbss, "botsort": fuzz_botsort, "deepocsort": fuzz_deepocsort, "ssort": fuzz_ssort}
out = {}
for name in WHICH:
    t0, ok = time.time(), 0
    for t in range(N):
        try:
            ok += bool(FUZZ[name](t, np.random.default_rng(9000 + t)))
        except _lib.TlkError as ex:                          # a capacity overflow on an extreme draw is reported, not a divergence
            print(f"CAPACITY {name} trial {t}: {ex}")
    s = STATS.get(name, {"rows": 0, "kalman_states": 0})
    out[name] = {"trials": N, "identical": ok, **s, "seconds": round(time.time() - t0, 1)}
    print(f"{name}: {ok}/{N} trials identical to the oracle (rows and Kalman states bit for bit); {s['rows']} rows, {s['kalman_states']} track states compared; {time.time() - t0:.0f} s", flush=True)
if LOAD is not None:
    LOAD[0].set(); LOAD[1].join(timeout=60)
    out["resnet_forwards_beside_the_run"] = LOAD[2][0]
    print("ResNet-50 bf16 forwards run on a side stream during the fuzz:", LOAD[2][0])
os.makedirs("gpurun_out", exist_ok=True)
json.dump(out, open("gpurun_out/fuzz_gpu%s.json" % ("_under_load" if LOAD is not None else ""), "w"), indent=1)
