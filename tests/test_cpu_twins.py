"""CPU: the `_cpu` twins of libtlk's minimum export list (include/tlk_cpu.h, oracle/src/cpu_twins.c).
  * every twin declared in tlk_cpu.h is exported by the ORACLE library and by it only -- libtlk.so has no CPU code;
  * every twin's parameter list is its libtlk counterpart's (same types in the same order: one function-pointer type serves both);
  * every twin returns what the orc_* restatement returns for the same inputs (the restatement is what the golden vectors pin)."""
import ctypes as C
import os
import re
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _decls(path):
    """{name: [normalised parameter types]} of every `int tlk_*(...)` prototype in a header"""
    txt = re.sub(r"/\*.*?\*/", " ", open(path).read(), flags=re.S)
    out = {}
    for m in re.finditer(r"\bint\s+(tlk_\w+)\s*\(([^;{}]*?)\)\s*;", txt, flags=re.S):
        params = []
        for prm in m.group(2).split(","):
            prm = " ".join(prm.split())
            if prm in ("void", ""):
                continue
            ty = re.sub(r"\b\w+$", "", prm).strip() if not prm.endswith("*") else prm        # drop the parameter name
            params.append(ty.replace(" *", "*").replace("* ", "*"))
        out[m.group(1)] = params
    return out


def test_every_twin_has_the_signature_of_its_libtlk_function_and_lives_in_the_oracle_library(orc):
    twins = _decls(os.path.join(ROOT, "include", "tlk_cpu.h"))
    twins = {k: v for k, v in twins.items() if k.endswith("_cpu")}
    main = _decls(os.path.join(ROOT, "include", "tlk.h"))
    assert len(twins) == 25
    need = ["tlk_ocsort_create", "tlk_ocsort_update", "tlk_ocsort_destroy", "tlk_bpbss_create", "tlk_bpbss_update", "tlk_bpbss_destroy",
            "tlk_iou_matrix_f64", "tlk_iou_ltwh_cost_f64", "tlk_oks_cost_f64", "tlk_partdist_f32", "tlk_cosine_gallery_min_f32",
            "tlk_kf7_predict_f64", "tlk_kf7_update_f64", "tlk_kf8_initiate_f64", "tlk_kf8_predict_f64", "tlk_kf8_project_f64",
            "tlk_kf8_update_f64", "tlk_kf8_gate_f64", "tlk_lsa_f64", "tlk_lsa_lapjv_limit_f64", "tlk_letterbox_u8",
            "tlk_roi_crop_resize_norm", "tlk_yolox_decode_nms"]                     # SURVEY.md 8(b), "minimum C-ABI export list"
    for name in need:
        assert name + "_cpu" in twins, name
    for name, params in twins.items():
        base = name[:-4]
        assert base in main, base
        want = [p.replace("tlk_ocsort*", "tlk_ocsort_cpu*").replace("tlk_bpbss*", "tlk_bpbss_cpu*") for p in main[base]]
        assert params == want, (name, params, want)
    exported = subprocess.run(["nm", "-D", "--defined-only", orc._LIB_PATH], capture_output=True, text=True, check=True).stdout
    exported = set(re.findall(r"\b(tlk_\w+_cpu)\b", exported))
    assert exported == set(twins)
    libtlk = os.path.join(ROOT, "tracklab_amd", "lib", "libtlk.so")
    if os.path.exists(libtlk):
        syms = subprocess.run(["nm", "-D", "--defined-only", libtlk], capture_output=True, text=True, check=True).stdout
        assert not re.findall(r"\b\w+_cpu\b", syms) and not re.findall(r"\borc_\w+", syms)      # the product has no CPU path


def _p(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


@pytest.fixture(scope="module")
def T(orc):
    L = orc.lib()
    for n in ("tlk_iou_matrix_f64_cpu", "tlk_lsa_f64_cpu", "tlk_kf8_gate_f64_cpu", "tlk_partdist_f32_cpu"):
        getattr(L, n).restype = C.c_int
    return L


def test_stateless_twins_equal_the_restatement(orc, T):
    rng = np.random.default_rng(0)
    b1 = np.concatenate([rng.uniform(0, 500, (37, 2)), rng.uniform(0, 500, (37, 2)) + 600], 1)
    b2 = np.concatenate([rng.uniform(0, 500, (23, 2)), rng.uniform(0, 500, (23, 2)) + 600], 1)
    for vi, var in enumerate(("iou", "giou", "diou", "ciou", "ct_dist")):
        out = np.empty((37, 23))
        assert T.tlk_iou_matrix_f64_cpu(vi, _p(b1), 37, _p(b2), 23, _p(out), None) == 0
        assert np.array_equal(out, orc.iou_matrix(b1, b2, var))
    assert T.tlk_iou_matrix_f64_cpu(7, _p(b1), 37, _p(b2), 23, _p(out), None) == -1 and T.tlk_iou_matrix_f64_cpu(0, None, 0, _p(b2), 23, None, None) == 0
    # batched assignment: rows sorted, -1 beyond the pair count, the infeasible / NaN codes of tlk_lsa_f64
    cost = rng.uniform(0, 1, (4, 9, 13))
    cost[2, :, :] = np.inf
    cost[3, 1, 2] = np.nan
    rows, cols, n = np.empty((4, 9), np.int32), np.empty((4, 9), np.int32), np.empty(4, np.int32)
    assert T.tlk_lsa_f64_cpu(_p(cost), 4, 9, 13, _p(rows), _p(cols), _p(n), None) == 0
    assert list(n) == [9, 9, -1, -2]
    for b in (0, 1):
        er, ec = orc.lsa(cost[b])
        assert np.array_equal(rows[b], er) and np.array_equal(cols[b], ec)
    x, y = np.empty((2, 9), np.int32), np.empty((2, 13), np.int32)
    T.tlk_lsa_lapjv_limit_f64_cpu.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_double, C.c_void_p, C.c_void_p, C.c_void_p]
    assert T.tlk_lsa_lapjv_limit_f64_cpu(_p(cost), 2, 9, 13, 0.4, _p(x), _p(y), None) == 0
    for b in (0, 1):
        ex, ey = orc.lapjv_limit(cost[b], 0.4)
        assert np.array_equal(x[b], ex) and np.array_equal(y[b], ey)
    # Kalman steps, batched = the single-filter restatement applied entry by entry
    meas = np.stack([rng.uniform(100, 900, 6), rng.uniform(100, 500, 6), rng.uniform(0.3, 0.6, 6), rng.uniform(80, 300, 6)], 1)
    mean, cov = np.empty((6, 8)), np.empty((6, 8, 8))
    assert T.tlk_kf8_initiate_f64_cpu(_p(meas), _p(mean), _p(cov), 6, None) == 0
    assert T.tlk_kf8_predict_f64_cpu(_p(mean), _p(cov), 6, None) == 0
    for i in range(6):
        m0, c0 = orc.kf8_initiate(meas[i])
        m1, c1 = orc.kf8_predict(m0, c0)
        assert np.array_equal(mean[i], m1) and np.array_equal(cov[i], c1)
    conf = rng.uniform(0.3, 1, 6)
    pm, pc = np.empty((6, 4)), np.empty((6, 4, 4))
    assert T.tlk_kf8_project_f64_cpu(_p(mean), _p(cov), _p(conf), _p(pm), _p(pc), 6, None) == 0
    gate = np.empty((6, 6))
    assert T.tlk_kf8_gate_f64_cpu(_p(mean), _p(cov), 6, _p(meas), 6, 1, _p(gate), None) == 0
    for i in range(6):
        a, b = orc.kf8_project(mean[i], cov[i], conf[i])
        assert np.array_equal(pm[i], a) and np.array_equal(pc[i], b)
        assert np.array_equal(gate[i], orc.kf8_gating(mean[i], cov[i], meas, True))
    m2, c2 = mean.copy(), cov.copy()
    assert T.tlk_kf8_update_f64_cpu(_p(m2), _p(c2), _p(meas), None, 6, None) == 0          # conf NULL = 0
    for i in range(6):
        a, b = orc.kf8_update(mean[i], cov[i], meas[i], 0.0)
        assert np.array_equal(m2[i], a) and np.array_equal(c2[i], b)
    x7 = rng.uniform(1, 100, (5, 7)); P7 = np.stack([np.eye(7) * (i + 1) for i in range(5)]); z = rng.uniform(1, 100, (5, 4))
    xa, Pa = x7.copy(), P7.copy()
    assert T.tlk_kf7_predict_f64_cpu(_p(xa), _p(Pa), 5, None) == 0 and T.tlk_kf7_update_f64_cpu(_p(xa), _p(Pa), _p(z), 5, None) == 0
    for i in range(5):
        a, b = orc.kf7_predict(x7[i], P7[i])
        a, b = orc.kf7_update(a, b, z[i])
        assert np.array_equal(xa[i], a) and np.array_equal(Pa[i], b)
    # motion and appearance costs
    tl = np.concatenate([rng.uniform(0, 800, (7, 2)), rng.uniform(20, 200, (7, 2))], 1); dl = np.concatenate([rng.uniform(0, 800, (9, 2)), rng.uniform(20, 200, (9, 2))], 1)
    out = np.empty((7, 9))
    assert T.tlk_iou_ltwh_cost_f64_cpu(_p(tl), 7, _p(dl), 9, _p(out), None) == 0 and np.array_equal(out, orc.iou_ltwh_cost(tl, dl))
    tk, dk = rng.uniform(0, 500, (7, 17, 3)), rng.uniform(0, 500, (9, 17, 3))
    tk[..., 2] = rng.uniform(0, 1, (7, 17)); dk[..., 2] = rng.uniform(0, 1, (9, 17))
    assert T.tlk_oks_cost_f64_cpu(_p(tk), 7, _p(dk), 9, _p(out), None) == 0 and np.array_equal(out, orc.oks_cost(tk, dk), equal_nan=True)
    q, g = rng.standard_normal((7, 6, 32)).astype(np.float32), rng.standard_normal((9, 6, 32)).astype(np.float32)
    qv, gv = (rng.random((7, 6)) > 0.3).astype(np.uint8), (rng.random((9, 6)) > 0.3).astype(np.uint8)
    assert T.tlk_partdist_f32_cpu(_p(q), _p(qv), 7, _p(g), _p(gv), 9, 6, 32, _p(out), None) == 0 and np.array_equal(out, orc.partdist(q, qv, g, gv))
    assert T.tlk_partdist_f32_cpu(_p(q), _p(qv), 7, _p(g), _p(gv), 9, 6, 30, _p(out), None) == -1          # libtlk's D % 16 rule
    gal = rng.standard_normal((20, 32)).astype(np.float32); offs = np.array([0, 3, 3, 10, 20], np.int32); dets = rng.standard_normal((9, 32)).astype(np.float32)
    o2 = np.empty((4, 9))
    assert T.tlk_cosine_gallery_min_f32_cpu(_p(gal), _p(offs), 4, 20, _p(dets), 9, 32, _p(o2), None) == 0
    assert np.array_equal(o2, orc.cosine_gallery_min(gal, offs, dets), equal_nan=True)


def test_image_twins_equal_the_restatement(orc, T):
    rng = np.random.default_rng(1)
    frames = rng.integers(0, 256, (2, 90, 160, 3), dtype=np.uint8)
    out = np.empty((2, 3, 64, 64), np.float32)
    ratio = C.c_double(0)
    assert T.tlk_letterbox_u8_cpu(_p(frames), 2, 90, 160, 64, 0, 0, _p(out), C.byref(ratio), None) == 0
    for b in range(2):
        e, r = orc.letterbox(frames[b], 64)
        assert np.array_equal(out[b], e) and ratio.value == r
    assert T.tlk_letterbox_u8_cpu(_p(frames), 2, 90, 160, 64, 1, 0, _p(out), None, None) == -5          # NHWC: libtlk's own layout, TLK_EUNSUPPORTED
    boxes = np.array([[[10.4, 5.5, 50.2, 60.7], [100.0, 20.0, 80.0, 80.0], [0, 0, 0, 0]], [[-5.0, -5.0, 30.0, 30.0], [0, 0, 0, 0], [0, 0, 0, 0]]], np.float32)
    counts = np.array([2, 1], np.int32)
    mean, std = np.array(orc.IMAGENET_MEAN, np.float32), np.array(orc.IMAGENET_STD, np.float32)
    crops = np.full((6, 3, 32, 16), 7.0, np.float32)
    assert T.tlk_roi_crop_resize_norm_cpu(_p(frames), 2, 90, 160, _p(boxes), _p(counts), 3, 32, 16, _p(mean), _p(std), 0, 0, _p(crops), None) == 0
    for b, i in ((0, 0), (0, 1), (1, 0)):
        ltrb = orc.ltwh_to_crop_ltrb(boxes[b, i].astype(np.float64), 160, 90)
        assert np.array_equal(crops[b * 3 + i], orc.crop_resize_norm(frames[b], ltrb, 32, 16)[0])
    assert np.all(crops[2] == 7.0) and np.all(crops[4:] == 7.0)                                      # slots >= counts[b] are not written
    from tracklab_amd.synth import SyntheticStream, synth_yolox_head as synth_head
    preds = np.stack([synth_head(rng, SyntheticStream(30 + b, 20, 1).step()["dets"][:, :4], dup=2) for b in range(2)])
    A = preds.shape[1]
    mo = 64
    ltwh, xyxy = np.zeros((2, mo, 4), np.float32), np.zeros((2, mo, 4), np.float32)
    sc, cl, cnt, trk = np.zeros((2, mo), np.float32), np.zeros((2, mo), np.int32), np.zeros(2, np.int32), np.zeros((2, mo, 7))
    T.tlk_yolox_decode_nms_cpu.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_float, C.c_float, C.c_float, C.c_int, C.c_int, C.c_int] + [C.c_void_p] * 6 + \
        [C.c_int64, C.c_double, C.c_void_p]
    ratio32 = float(np.float32(640 / 1920))
    assert T.tlk_yolox_decode_nms_cpu(_p(preds), 2, 640, preds.shape[2] - 5, ratio32, 0.45, 0.7, 1920, 1080, mo, _p(ltwh), _p(xyxy), _p(sc), _p(cl), _p(cnt), _p(trk),
                                      1000, 1.0, None) == 0
    assert A == 8400
    for b in range(2):
        eb, es, ec = orc.yolox_postprocess(preds[b], 640, ratio32)
        n = cnt[b]
        assert n == len(eb) and n >= 15
        assert np.array_equal(xyxy[b, :n], eb) and np.array_equal(sc[b, :n], es) and np.array_equal(cl[b, :n], ec)
        l = np.maximum(0, np.minimum(eb[:, 0], 1918)).astype(np.float32); t = np.maximum(0, np.minimum(eb[:, 1], 1078)).astype(np.float32)
        r = np.maximum(1, np.minimum(eb[:, 2], 1919)).astype(np.float32); bt = np.maximum(1, np.minimum(eb[:, 3], 1079)).astype(np.float32)
        assert np.array_equal(ltwh[b, :n], np.stack([l, t, r - l, bt - t], 1))
        assert np.array_equal(trk[b, :n, 6], 1000 + b * mo + np.arange(n)) and np.all(trk[b, :n, 4] == 1.0)
        assert np.array_equal(trk[b, :n, 2], (l + (r - l)).astype(np.float64))
    assert T.tlk_yolox_decode_nms_cpu(_p(preds), 2, 640, preds.shape[2] - 5, ratio32, 0.45, 0.7, 1920, 1080, 4, _p(ltwh), _p(xyxy), _p(sc), _p(cl), _p(cnt), None,
                                      0, 1.0, None) == 0 and list(cnt) == [-3, -3]                  # TLK_ECAPACITY per frame, like libtlk


def test_tracker_bank_twins_equal_the_restatement(orc, T):
    from tracklab_amd import _lib
    from tracklab_amd.synth import SyntheticStream
    # OC-SORT, wrapper mode (OCSORT.process: empty frames skipped, conf > min_confidence)
    p = _lib.OcsortParams(det_thresh=0.3, iou_threshold=0.3, inertia=0.2, min_confidence=0.4, max_age=30, min_hits=3, delta_t=3, asso_func=1,
                          use_byte=0, wrapper_mode=1, max_tracks=256, max_dets=128)
    h = C.c_void_p()
    T.tlk_ocsort_create_cpu.argtypes = [C.POINTER(_lib.OcsortParams), C.c_int, C.c_int, C.POINTER(C.c_void_p)]
    T.tlk_ocsort_update_cpu.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.POINTER(C.c_int)]
    T.tlk_ocsort_destroy_cpu.argtypes = [C.c_void_p]
    T.tlk_ocsort_reset_cpu.argtypes = [C.c_void_p, C.c_int]
    assert T.tlk_ocsort_create_cpu(C.byref(p), 2, 0, C.byref(h)) == 0
    refs = [orc.OCSort(0.3, 30, 3, 0.3, 3, "giou", 0.2, False) for _ in range(2)]
    streams = [SyntheticStream(11 + s, 30, 1) for s in range(2)]
    total = 0
    for f in range(40):
        for s in range(2):
            d = streams[s].step()["dets"]
            dets = np.zeros((0, 7)) if f % 9 == 4 else np.ascontiguousarray(np.concatenate([d[:, :4], d[:, 4:5], np.zeros((len(d), 1)), np.arange(len(d))[:, None] + 1000 * f], 1))
            out, n = np.empty((256, 8)), C.c_int(0)
            assert T.tlk_ocsort_update_cpu(h, s, _p(dets), len(dets), _p(out), 256, C.byref(n)) == 0
            exp = orc.ocsort_wrapper_step(refs[s], dets, 0.4)
            assert n.value == len(exp) and np.array_equal(out[:n.value], exp)
            total += n.value
    assert total > 1000
    assert T.tlk_ocsort_reset_cpu(h, 1) == 0 and T.tlk_ocsort_update_cpu(h, 5, None, 0, None, 0, C.byref(n)) == -1
    assert T.tlk_ocsort_destroy_cpu(h) == 0
    # BPBReID-StrongSORT
    bp = _lib.BpbssParams(ema_alpha=0.9, mc_lambda=0.995, max_dist=0.2, max_iou_distance=0.7, min_bbox_confidence=0.3, gating_thres_factor=1.5, w_kfgd=1, w_reid=1,
                          w_st=1, max_age=30, n_init=2, only_position_for_kf_gating=0, max_kalman_prediction_without_update=7, matching_strategy=0, wrapper_mode=1,
                          parts=6, dim=32, max_tracks=256, max_dets=128, motion_criterium=0, reserved_=0, max_oks_distance=0.7)
    T.tlk_bpbss_create_cpu.argtypes = [C.POINTER(_lib.BpbssParams), C.c_int, C.c_int, C.POINTER(C.c_void_p)]
    T.tlk_bpbss_update_cpu.argtypes = [C.c_void_p, C.c_int] + [C.c_void_p] * 6 + [C.c_int, C.c_void_p, C.c_int, C.POINTER(C.c_int)]
    T.tlk_bpbss_destroy_cpu.argtypes = [C.c_void_p]
    hb = C.c_void_p()
    assert T.tlk_bpbss_create_cpu(C.byref(bp), 1, 0, C.byref(hb)) == 0
    ref = orc.StrongSORT(6, 32, ema_alpha=0.9, mc_lambda=0.995, max_dist=0.2, max_iou_distance=0.7, min_bbox_confidence=0.3, gating_thres_factor=1.5, max_age=30, n_init=2)
    st = SyntheticStream(5, 25, 1, parts=6, dim=32, with_embeddings=True)
    rows_total = 0
    for f in range(30):
        fr = st.step()
        d = fr["dets"]
        n = len(d)
        ids = np.arange(n, dtype=np.int64) + 100 * f
        ltwh = np.ascontiguousarray(np.stack([d[:, 0], d[:, 1], d[:, 2] - d[:, 0], d[:, 3] - d[:, 1]], 1))
        emb, vis, conf = np.ascontiguousarray(fr["embeddings"], dtype=np.float32), np.ascontiguousarray(fr["visibility"], dtype=np.uint8), np.ascontiguousarray(d[:, 4])
        rows, cnt = np.zeros(128, _lib.BPBSS_ROW), C.c_int(0)
        assert T.tlk_bpbss_update_cpu(hb, 0, _p(ids), _p(ltwh), _p(emb), _p(vis), _p(conf), None, n, _p(rows), 128, C.byref(cnt)) == 0
        exp = ref.update(ids, ltwh, emb, vis, conf)
        assert cnt.value == len(exp)
        for k in ("det_id", "track_id", "kf_ltwh", "hits", "age", "tsu", "state", "matched_name"):
            assert np.array_equal(rows[k][:cnt.value], exp[k]), (f, k)
        rows_total += cnt.value
    assert rows_total > 300
    assert T.tlk_bpbss_destroy_cpu(hb) == 0
