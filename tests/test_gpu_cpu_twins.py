"""GPU: libtlk and its `_cpu` twins (include/tlk_cpu.h; the twins live in the oracle library) called through ONE argument list per function --
device buffers on one side, host buffers on the other, everything else identical -- on the minimum export list of SURVEY.md 8(b).  Integer and
fp64 results are bit-identical (the Kalman / IoU / assignment contracts of tests/test_gpu_kernels.py), the fp32 MFMA contraction and the device
exp keep the tolerances of their own parity tests."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _both(name, args, outs, orc):
    """call libtlk's `name` on device copies and `name`_cpu on the host arrays; `args` is the shared argument list (numpy arrays stand for
    pointers), `outs` the indices of the output arrays; returns [(device result, host result), ...] per output"""
    import torch
    from tracklab_amd import _lib
    L, T = _lib.lib(), orc.lib()
    _lib._bind_image(L); _lib._bind_bpbss(L); _lib._bind_kf(L)
    dev = [torch.from_numpy(a).cuda() if isinstance(a, np.ndarray) else a for a in args]
    fn_d, fn_h = getattr(L, name), getattr(T, name + "_cpu")
    fn_h.argtypes = fn_d.argtypes
    fn_h.restype = C.c_int
    rc_d = fn_d(*[d.data_ptr() if hasattr(d, "data_ptr") else d for d in dev])
    torch.cuda.synchronize()
    rc_h = fn_h(*[a.ctypes.data if isinstance(a, np.ndarray) else a for a in args])
    assert rc_d == rc_h == 0, (name, rc_d, rc_h)
    return [(dev[i].cpu().numpy(), args[i]) for i in outs]


def test_stateless_kernels_and_their_twins_through_one_argument_list(orc):
    rng = np.random.default_rng(3)
    b1 = np.concatenate([rng.uniform(0, 900, (100, 2)), rng.uniform(0, 900, (100, 2)) + 950], 1)
    b2 = np.concatenate([rng.uniform(0, 900, (90, 2)), rng.uniform(0, 900, (90, 2)) + 950], 1)
    for v in range(5):
        (d, h), = _both("tlk_iou_matrix_f64", [v, b1, 100, b2, 90, np.empty((100, 90)), None], [5], orc)
        if v < 3:
            assert np.array_equal(d, h)
        else:
            np.testing.assert_allclose(d, h, rtol=1e-12, atol=1e-14)                 # ciou / ct_dist: device atan / sqrt (tests/test_gpu_kernels.py)
    cost = rng.uniform(0, 1, (6, 100, 100))
    cost[4, :, 7] = np.inf; cost[4, :, 8] = np.inf; cost[4, 3:, :] = np.inf            # infeasible
    res = _both("tlk_lsa_f64", [cost, 6, 100, 100, np.empty((6, 100), np.int32), np.empty((6, 100), np.int32), np.empty(6, np.int32), None], [4, 5, 6], orc)
    n = res[2][1]
    assert np.array_equal(res[2][0], n) and n[4] == -1 and (n[[0, 1, 2, 3, 5]] == 100).all()
    for b in (0, 1, 2, 3, 5):
        assert np.array_equal(res[0][0][b], res[0][1][b]) and np.array_equal(res[1][0][b], res[1][1][b])
    meas = np.stack([rng.uniform(100, 1800, 64), rng.uniform(100, 900, 64), rng.uniform(0.3, 0.6, 64), rng.uniform(80, 300, 64)], 1)
    (m_d, m_h), (c_d, c_h) = _both("tlk_kf8_initiate_f64", [meas, np.empty((64, 8)), np.empty((64, 8, 8)), 64, None], [1, 2], orc)
    assert np.array_equal(m_d, m_h) and np.array_equal(c_d, c_h)
    (m_d, m_h), (c_d, c_h) = _both("tlk_kf8_predict_f64", [m_h, c_h, 64, None], [0, 1], orc)
    assert np.array_equal(m_d, m_h) and np.array_equal(c_d, c_h)
    for only_pos in (0, 1):
        (g_d, g_h), = _both("tlk_kf8_gate_f64", [m_h, c_h, 64, meas, 64, only_pos, np.empty((64, 64)), None], [6], orc)
        assert np.array_equal(g_d, g_h)
    conf = rng.uniform(0.2, 1, 64)
    (m_d, m_h2), (c_d, c_h2) = _both("tlk_kf8_update_f64", [m_h.copy(), c_h.copy(), meas[::-1].copy(), conf, 64, None], [0, 1], orc)
    assert np.array_equal(m_d, m_h2) and np.array_equal(c_d, c_h2)
    tl = np.concatenate([rng.uniform(0, 1700, (100, 2)), rng.uniform(30, 300, (100, 2))], 1)
    dl = tl[rng.permutation(100)[:90]] + rng.normal(0, 3, (90, 4))
    (d, h), = _both("tlk_iou_ltwh_cost_f64", [tl, 100, dl, 90, np.empty((100, 90)), None], [4], orc)
    assert np.array_equal(d, h)
    tk, dk = rng.uniform(0, 900, (40, 17, 3)), rng.uniform(0, 900, (30, 17, 3))
    tk[..., 2] = rng.uniform(0, 1, (40, 17)); dk[..., 2] = rng.uniform(0, 1, (30, 17))
    (d, h), = _both("tlk_oks_cost_f64", [tk, 40, dk, 30, np.empty((40, 30)), None], [4], orc)
    np.testing.assert_allclose(d, h, rtol=1e-13, atol=1e-16, equal_nan=True)
    proto = rng.normal(0, 1, (100, 6, 256)).astype(np.float32)
    q, g = (proto + 0.1 * rng.normal(0, 1, proto.shape)).astype(np.float32), (proto[:90] + 0.1 * rng.normal(0, 1, (90, 6, 256))).astype(np.float32)
    qv, gv = (rng.random((100, 6)) < 0.8).astype(np.uint8), (rng.random((90, 6)) < 0.8).astype(np.uint8)
    (d, h), = _both("tlk_partdist_f32", [q, qv, 100, g, gv, 90, 6, 256, np.empty((100, 90)), None], [8], orc)
    np.testing.assert_allclose(d, h, rtol=1e-5, atol=1e-5)                            # fp32 MFMA contraction (tests/test_gpu_bpbss.py)


def test_detector_post_processing_and_its_twin(orc):
    from tracklab_amd.synth import SyntheticStream, synth_yolox_head
    rng = np.random.default_rng(5)
    preds = np.stack([synth_yolox_head(rng, SyntheticStream(60 + b, 100, 1).step()["dets"][:, :4], dup=2) for b in range(3)])
    mo = 256
    args = [preds, 3, 640, 1, float(np.float32(640 / 1920)), 0.45, 0.7, 1920, 1080, mo, np.zeros((3, mo, 4), np.float32), np.zeros((3, mo, 4), np.float32),
            np.zeros((3, mo), np.float32), np.zeros((3, mo), np.int32), np.zeros(3, np.int32), np.zeros((3, mo, 7)), 5000, 1.0, None]
    (l_d, l_h), (x_d, x_h), (s_d, s_h), (c_d, c_h), (n_d, n_h), (t_d, t_h) = _both("tlk_yolox_decode_nms", args, [10, 11, 12, 13, 14, 15], orc)
    assert np.array_equal(n_d, n_h) and (n_h >= 80).all()
    for b in range(3):
        n = n_h[b]
        assert np.array_equal(s_d[b, :n], s_h[b, :n]) and np.array_equal(c_d[b, :n], c_h[b, :n])
        np.testing.assert_allclose(x_d[b, :n], x_h[b, :n], rtol=2e-6, atol=1e-4)     # device expf: 1-2 ulp (tests/test_gpu_image.py)
        np.testing.assert_allclose(l_d[b, :n], l_h[b, :n], rtol=2e-6, atol=2e-4)
        np.testing.assert_allclose(t_d[b, :n, :4], t_h[b, :n, :4], rtol=2e-6, atol=3e-4)
        assert np.array_equal(t_d[b, :n, 4:], t_h[b, :n, 4:])                           # conf 1.0, category, detection ids


def test_tracker_banks_and_their_twins_step_frame_by_frame_to_the_same_rows(orc):
    """the host-buffer entry points (tlk_*_update: what the TrackLab Module calls) take the SAME arguments on both sides"""
    from tracklab_amd import _lib
    from tracklab_amd.synth import SyntheticStream
    L, T = _lib.lib(), orc.lib()
    _lib._bind_bpbss(L)
    p = _lib.OcsortParams(det_thresh=0.3, iou_threshold=0.3, inertia=0.2, min_confidence=0.4, max_age=30, min_hits=3, delta_t=3, asso_func=0,
                          use_byte=0, wrapper_mode=1, max_tracks=512, max_dets=256)
    for lib_, suffix in ((L, ""), (T, "_cpu")):
        for n in ("create", "update", "destroy"):
            getattr(lib_, f"tlk_ocsort_{n}{suffix}").argtypes = getattr(L, f"tlk_ocsort_{n}").argtypes
            getattr(lib_, f"tlk_bpbss_{n}{suffix}").argtypes = getattr(L, f"tlk_bpbss_{n}").argtypes
    hd, hh = C.c_void_p(), C.c_void_p()
    assert L.tlk_ocsort_create(C.byref(p), 1, 0, C.byref(hd)) == 0 and T.tlk_ocsort_create_cpu(C.byref(p), 1, 0, C.byref(hh)) == 0
    st = SyntheticStream(21, 100, 1, low_conf_frac=0.1)
    dp = C.POINTER(C.c_double)
    rows = 0
    for f in range(60):
        d = np.zeros((0, 7)) if f == 17 else np.ascontiguousarray(st.step()["dets"])
        od, oh, nd, nh = np.empty((512, 8)), np.empty((512, 8)), C.c_int(0), C.c_int(0)
        a = (0, d.ctypes.data_as(dp), len(d))
        assert L.tlk_ocsort_update(hd, *a, od.ctypes.data_as(dp), 512, C.byref(nd)) == 0
        assert T.tlk_ocsort_update_cpu(hh, *a, oh.ctypes.data_as(dp), 512, C.byref(nh)) == 0
        assert nd.value == nh.value and np.array_equal(od[:nd.value], oh[:nh.value]), f
        rows += nd.value
    assert rows > 4000
    assert L.tlk_ocsort_destroy(hd) == 0 and T.tlk_ocsort_destroy_cpu(hh) == 0
    bp = _lib.BpbssParams(ema_alpha=0.9, mc_lambda=0.995, max_dist=0.2, max_iou_distance=0.7, min_bbox_confidence=0.3, gating_thres_factor=1.5, w_kfgd=1, w_reid=1,
                          w_st=1, max_age=30, n_init=3, only_position_for_kf_gating=0, max_kalman_prediction_without_update=7, matching_strategy=0, wrapper_mode=1,
                          parts=6, dim=64, max_tracks=512, max_dets=256, motion_criterium=0, reserved_=0, max_oks_distance=0.7)
    assert L.tlk_bpbss_create(C.byref(bp), 1, 0, C.byref(hd)) == 0 and T.tlk_bpbss_create_cpu(C.byref(bp), 1, 0, C.byref(hh)) == 0
    st = SyntheticStream(8, 100, 1, parts=6, dim=64, with_embeddings=True)
    rows = 0
    for f in range(40):
        fr = st.step()
        d = fr["dets"]
        n = len(d)
        ids = np.arange(n, dtype=np.int64) + 1000 * f
        ltwh = np.ascontiguousarray(np.stack([d[:, 0], d[:, 1], d[:, 2] - d[:, 0], d[:, 3] - d[:, 1]], 1))
        emb, vis, conf = np.ascontiguousarray(fr["embeddings"], dtype=np.float32), np.ascontiguousarray(fr["visibility"], dtype=np.uint8), np.ascontiguousarray(d[:, 4])
        rd, rh, nd, nh = np.zeros(256, _lib.BPBSS_ROW), np.zeros(256, _lib.BPBSS_ROW), C.c_int(0), C.c_int(0)
        a = (0, ids.ctypes.data, ltwh.ctypes.data, emb.ctypes.data, vis.ctypes.data, conf.ctypes.data, None, n)
        assert L.tlk_bpbss_update(hd, *a, rd.ctypes.data, 256, C.byref(nd)) == 0
        assert T.tlk_bpbss_update_cpu(hh, *a, rh.ctypes.data, 256, C.byref(nh)) == 0
        assert nd.value == nh.value, f
        k = nd.value
        for key in ("det_id", "track_id", "hits", "age", "tsu", "state", "matched_name", "pred_valid"):
            assert np.array_equal(rd[key][:k], rh[key][:k]), (f, key)
        assert np.array_equal(rd["kf_ltwh"][:k], rh["kf_ltwh"][:k]), f                 # library operation order (oracle/src/lapack_order.h): bit-exact
        rows += k
    assert rows > 2500
    assert L.tlk_bpbss_destroy(hd) == 0 and T.tlk_bpbss_destroy_cpu(hh) == 0
