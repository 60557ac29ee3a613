"""-m gpu: stateless kernels through the C ABI vs the oracle / scipy / golden vectors."""
import json
import os
from ctypes import c_double as C_double, c_int as C_int, c_void_p as C_void_p

import numpy as np
import pytest

from conftest import GOLDEN

pytestmark = pytest.mark.gpu


def _iou_gpu(b1, b2, variant):
    import torch
    from tracklab_amd._lib import ASSO, check, lib
    d1 = torch.from_numpy(np.ascontiguousarray(b1[:, :4])).cuda()
    d2 = torch.from_numpy(np.ascontiguousarray(b2[:, :4])).cuda()
    out = torch.empty((len(b1), len(b2)), dtype=torch.float64, device="cuda")
    torch.cuda.synchronize()
    check(lib().tlk_iou_matrix_f64(ASSO[variant], d1.data_ptr(), len(b1), d2.data_ptr(), len(b2), out.data_ptr(), None))
    torch.cuda.synchronize()
    return out.cpu().numpy()


def test_iou_family_golden_bit_exact():
    g = np.load(os.path.join(GOLDEN, "iou_family.npz"))
    for tag in "abcd":
        b1, b2 = g[f"{tag}_b1"], g[f"{tag}_b2"]
        for fn, var in (("iou_batch", "iou"), ("giou_batch", "giou"), ("diou_batch", "diou"),
                        ("ciou_batch", "ciou"), ("ct_dist", "ct_dist")):
            got = _iou_gpu(b1, b2, var)
            exp = g[f"{tag}_{fn}"]
            if var in ("iou", "giou", "diou"):
                np.testing.assert_array_equal(got, exp, err_msg=f"{tag} {fn}")
            else:       # atan / global-max rescale: not the same libm
                np.testing.assert_allclose(got, exp, rtol=1e-12, atol=1e-14, err_msg=f"{tag} {fn}")


def _lsa_gpu(costs):
    import torch
    from tracklab_amd._lib import check, lib
    costs = np.ascontiguousarray(costs, dtype=np.float64)
    B, nr, nc = costs.shape
    k = min(nr, nc)
    d = torch.from_numpy(costs).cuda()
    rows = torch.full((B, max(k, 1)), -7, dtype=torch.int32, device="cuda")
    cols = torch.full((B, max(k, 1)), -7, dtype=torch.int32, device="cuda")
    npairs = torch.full((B,), -9, dtype=torch.int32, device="cuda")
    torch.cuda.synchronize()
    check(lib().tlk_lsa_f64(d.data_ptr(), B, nr, nc, rows.data_ptr(), cols.data_ptr(), npairs.data_ptr(), None))
    torch.cuda.synchronize()
    return rows.cpu().numpy(), cols.cpu().numpy(), npairs.cpu().numpy()


def test_lsa_golden_and_known_answers():
    g = np.load(os.path.join(GOLDEN, "lsa_cases.npz"))
    for i in range(int(g["n_cases"])):
        c = g[f"c{i}_cost"]
        r, cc, n = _lsa_gpu(c[None])
        assert n[0] == len(g[f"c{i}_rows"]), f"case {i}"
        np.testing.assert_array_equal(r[0, :n[0]], g[f"c{i}_rows"], err_msg=f"case {i}")
        np.testing.assert_array_equal(cc[0, :n[0]], g[f"c{i}_cols"], err_msg=f"case {i}")
    kat = json.load(open(os.path.join(GOLDEN, "lsa_kat.json")))
    for case in kat["cases"]:
        r, cc, n = _lsa_gpu(np.array(case["cost"], dtype=float)[None])
        assert list(r[0, :n[0]]) == case["rows"] and list(cc[0, :n[0]]) == case["cols"]


def test_lsa_bounded_walk_trips_loudly_instead_of_spinning():
    """r05: every data-dependent loop of the device-side solver is bounded; the augmenting-path walk gives up with
    n_pairs = -3 (TLK_EINTERNAL in the tracker banks) when it exceeds its bound.  The natural bound (number of rows) cannot trip on a
    consistent state, so the test hook lowers it: with a cap of 1 hop every problem whose augmentation re-assigns more than one row must
    report -3 -- and return, not hang -- while the others still give scipy's answer; cap 0 restores the solver."""
    from scipy.optimize import linear_sum_assignment
    from tracklab_amd._lib import check, lib
    rng = np.random.default_rng(5)
    costs = rng.uniform(0, 1, (32, 100, 100))
    # (600 x 600: the LDS-array solver tier, 100 x 100: the register tier)
    big = rng.uniform(0, 1, (4, 600, 600))
    try:
        check(lib().tlk_debug_lsa_hop_limit(1))
        for batch in (costs, big):
            r, c, n = _lsa_gpu(batch)
            assert (n == -3).any(), "no augmenting path longer than one hop in random problems?"
            for b in np.nonzero(n >= 0)[0]:
                rr, cc = linear_sum_assignment(batch[b])
                np.testing.assert_array_equal(c[b, :n[b]], cc)
    finally:
        check(lib().tlk_debug_lsa_hop_limit(0))
    r, c, n = _lsa_gpu(costs)
    assert (n == 100).all()
    for b in range(len(costs)):
        rr, cc = linear_sum_assignment(costs[b])
        np.testing.assert_array_equal(c[b], cc)
    with pytest.raises(Exception):
        check(lib().tlk_debug_lsa_hop_limit(-1))


@pytest.mark.parametrize("shape", [(100, 100), (100, 130), (130, 100), (7, 200), (256, 256)])
def test_lsa_batched_vs_scipy(shape):
    from scipy.optimize import linear_sum_assignment
    rng = np.random.default_rng(shape[0] * 1000 + shape[1])
    B = 24
    costs = rng.uniform(0, 1, (B,) + shape)
    costs[1::4] = np.round(costs[1::4] * 4)                       # integer ties
    costs[2::4][costs[2::4] > 0.3] = 0.3 + 1e-5                    # clamped like linear_assignment.py:55
    costs[3::4] *= -1
    r, c, n = _lsa_gpu(costs)
    for b in range(B):
        er, ec = linear_sum_assignment(costs[b])
        assert n[b] == len(er)
        np.testing.assert_array_equal(r[b, :n[b]], er, err_msg=f"batch {b}")
        np.testing.assert_array_equal(c[b, :n[b]], ec, err_msg=f"batch {b}")


def test_lsa_invalid_inputs_report_like_scipy():
    c = np.zeros((2, 3, 3))
    c[0, 1, 1] = np.nan
    c[1] = np.inf
    r, cc, n = _lsa_gpu(c)
    assert n[0] == -2 and n[1] == -1


def test_cosine_gallery_mfma_matches_reference_golden_and_oracle(orc):
    """tlk_cosine_gallery_min_f32 vs strong_sort NearestNeighborDistanceMetric('cosine').distance (golden by import)."""
    import torch
    from tracklab_amd import _lib
    g = np.load(os.path.join(GOLDEN, "cosine_gallery.npz"))
    for c in range(int(g["n_cases"])):
        gal, offs, dets = g[f"c{c}_gallery"], g[f"c{c}_offsets"], g[f"c{c}_dets"]
        out = _lib.cosine_gallery_min(torch.from_numpy(gal).cuda(), torch.from_numpy(offs).cuda(), torch.from_numpy(dets).cuda())
        torch.cuda.synchronize()
        np.testing.assert_allclose(out.cpu().numpy(), g[f"c{c}_cost"], rtol=0, atol=3e-6, err_msg=f"case {c} vs reference")
        np.testing.assert_allclose(out.cpu().numpy(), orc.cosine_gallery_min(gal, offs, dets), rtol=0, atol=3e-6)


# ------------------------------------------------------------------------------------------------ stateless KF / motion costs
def _cuda(a):
    import torch
    return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float64)).cuda()


def test_kf7_stateless_golden_and_oracle(orc):
    from test_oracle_motion import kf7_chain_steps
    from tracklab_amd import _lib
    g = np.load(os.path.join(GOLDEN, "kf7_cases.npz"))
    steps = list(kf7_chain_steps(g))
    x = _cuda(np.stack([s[0] for s in steps])); P = _cuda(np.stack([s[1] for s in steps])); z = _cuda(np.stack([s[2] for s in steps]))
    _lib.kf7_predict_(x, P)
    xp, Pp = x.cpu().numpy().copy(), P.cpu().numpy().copy()
    _lib.kf7_update_(x, P, z)
    xu, Pu = x.cpu().numpy(), P.cpu().numpy()
    for i, (x0, P0, zi, x1, P1) in enumerate(steps):
        ox, oP = orc.kf7_predict(x0, P0)
        np.testing.assert_array_equal(xp[i], ox); np.testing.assert_array_equal(Pp[i], oP)          # bit-exact vs the oracle
        ox, oP = orc.kf7_update(ox, oP, zi)
        np.testing.assert_array_equal(xu[i], ox); np.testing.assert_array_equal(Pu[i], oP)
        np.testing.assert_allclose(xu[i], x1, rtol=1e-10, atol=1e-10)                                # and the reference's states
        np.testing.assert_allclose(Pu[i], P1, rtol=1e-9, atol=1e-9)


def test_kf8_stateless_golden_and_oracle(orc):
    from tracklab_amd import _lib
    g = np.load(os.path.join(GOLDEN, "kf8_cases.npz"))
    n = 32
    mean, cov = _lib.kf8_initiate(_cuda(g["meas"]))
    np.testing.assert_array_equal(mean.cpu().numpy(), g["init_mean"])
    np.testing.assert_array_equal(cov.cpu().numpy(), g["init_cov"])
    # case i is predicted i % 4 + 1 times: run 4 rounds, freezing the cases that are done
    m_np, c_np = mean.cpu().numpy().copy(), cov.cpu().numpy().copy()
    for rnd in range(4):
        sel = np.array([i for i in range(n) if i % 4 + 1 > rnd])
        ms, cs = _cuda(m_np[sel]), _cuda(c_np[sel])
        _lib.kf8_predict_(ms, cs)
        m_np[sel], c_np[sel] = ms.cpu().numpy(), cs.cpu().numpy()
    np.testing.assert_array_equal(m_np, g["pred_mean"])
    np.testing.assert_array_equal(c_np, g["pred_cov"])
    mean, cov, conf = _cuda(m_np), _cuda(c_np), _cuda(g["conf"])
    pm, pc = _lib.kf8_project(mean, cov, conf)
    np.testing.assert_array_equal(pm.cpu().numpy(), g["proj_mean"])
    np.testing.assert_array_equal(pc.cpu().numpy(), g["proj_cov"])
    z = np.stack([g[f"z{i}"] for i in range(n)])
    for only_pos, key in ((False, "gate4"), (True, "gate2")):
        gate = _lib.kf8_gate(mean, cov, _cuda(z), only_pos).cpu().numpy()          # (32 filters) x (32 measurements)
        for i in range(n):
            assert gate[i, i] == g[key][i][i % 50]        # bit-exact since r03 (LAPACK operation order, tlk_strongsort_common.hpp); both calls have >= 2 measurements
            np.testing.assert_array_equal(gate[i], orc.kf8_gating(m_np[i], c_np[i], z, only_pos))
        one = _lib.kf8_gate(mean, cov, _cuda(z[:1]), only_pos).cpu().numpy()        # ONE measurement: the library's trsv path (division, dot form)
        for i in range(n):
            np.testing.assert_array_equal(one[i], orc.kf8_gating(m_np[i], c_np[i], z[:1], only_pos))
    _lib.kf8_update_(mean, cov, _cuda(z), conf)
    mu, cu = mean.cpu().numpy(), cov.cpu().numpy()
    np.testing.assert_array_equal(mu, g["upd_mean"])           # bit-exact against the reference since r03
    np.testing.assert_array_equal(cu, g["upd_cov"])
    for i in range(n):
        om, oc = orc.kf8_update(m_np[i], c_np[i], z[i], g["conf"][i])
        np.testing.assert_array_equal(mu[i], om); np.testing.assert_array_equal(cu[i], oc)


def test_motion_costs_golden_and_oracle(orc):
    from tracklab_amd import _lib
    g = np.load(os.path.join(GOLDEN, "motion_costs.npz"))
    for c in range(int(g["n_cases"])):
        iou_c = _lib.iou_ltwh_cost(_cuda(g[f"c{c}_trk_ltwh"]), _cuda(g[f"c{c}_det_ltwh"])).cpu().numpy()
        np.testing.assert_array_equal(iou_c, g[f"c{c}_iou_cost"])
        oks_c = _lib.oks_cost(_cuda(g[f"c{c}_trk_kps"]), _cuda(g[f"c{c}_det_kps"])).cpu().numpy()
        np.testing.assert_allclose(oks_c, g[f"c{c}_oks_cost"], rtol=1e-12, atol=1e-15, equal_nan=True)
        np.testing.assert_allclose(oks_c, orc.oks_cost(g[f"c{c}_trk_kps"], g[f"c{c}_det_kps"]), rtol=1e-13, atol=1e-16, equal_nan=True)


def test_stateless_entry_points_reject_bad_arguments():
    import torch
    from tracklab_amd._lib import TlkError, check, lib, _bind_kf
    L = lib(); _bind_kf(L)
    with pytest.raises(TlkError):
        check(L.tlk_kf7_predict_f64(None, None, 4, None))
    with pytest.raises(TlkError):
        check(L.tlk_kf8_gate_f64(None, None, 3, None, 3, 0, None, None))
    with pytest.raises(TlkError):
        check(L.tlk_oks_cost_f64(None, -1, None, 2, None, None))
    check(L.tlk_iou_ltwh_cost_f64(None, 0, None, 5, None, None))        # empty problem is a no-op


def test_lapjv_cost_limit_matches_oracle(orc):
    import torch
    from tracklab_amd._lib import check, lib
    L = lib()
    L.tlk_lsa_lapjv_limit_f64.argtypes = [C_void_p, C_int, C_int, C_int, C_double, C_void_p, C_void_p, C_void_p]
    rng = np.random.default_rng(9)
    for nr, nc, limit, batch in [(30, 45, 0.3, 5), (100, 100, 0.8, 3), (7, 1, 0.5, 2), (1, 9, 2.0, 4), (60, 20, 0.05, 2)]:
        cost = rng.uniform(0, 1, (batch, nr, nc))
        d = torch.from_numpy(cost).cuda()
        x = torch.empty((batch, nr), dtype=torch.int32, device="cuda"); y = torch.empty((batch, nc), dtype=torch.int32, device="cuda")
        check(L.tlk_lsa_lapjv_limit_f64(d.data_ptr(), batch, nr, nc, limit, x.data_ptr(), y.data_ptr(), None))
        torch.cuda.synchronize()
        for b in range(batch):
            ex, ey = orc.lapjv_limit(cost[b], limit)
            np.testing.assert_array_equal(x[b].cpu().numpy(), ex); np.testing.assert_array_equal(y[b].cpu().numpy(), ey)
    y = torch.zeros((2, 4), dtype=torch.int32, device="cuda")
    check(L.tlk_lsa_lapjv_limit_f64(None, 2, 0, 4, 0.5, None, y.data_ptr(), None))         # no rows: every column unmatched
    torch.cuda.synchronize()
    assert (y.cpu().numpy() == -1).all()


@pytest.mark.parametrize("dtype_name", ["float16", "bfloat16"])
@pytest.mark.parametrize("act,residual", [("relu", True), ("relu", False), ("silu", False), (None, False), (None, True)])
def test_fused_gemm_bias_act_matches_fp32_reference(dtype_name, act, residual):
    """tlk_gemm_bias_act (one tuned hipBLASLt call: GEMM + bias + activation + residual) vs the same op in fp32 torch.
    Tolerance = one rounding of the output to the 16-bit type plus fp32 accumulation-order noise."""
    import torch
    import torch.nn.functional as F
    from tracklab_amd import _lib
    dt = getattr(torch, dtype_name)
    g = torch.Generator(device="cuda").manual_seed(3)
    for M, K, N in [(4096, 64, 256), (1000, 256, 64), (77, 128, 512)]:
        x = torch.randn(M, K, device="cuda", generator=g).to(dt)
        w = (torch.randn(N, K, device="cuda", generator=g) * 0.1).to(dt)
        b = torch.randn(N, device="cuda", generator=g).to(dt)
        r = torch.randn(M, N, device="cuda", generator=g).to(dt) if residual else None
        out = _lib.gemm_bias_act(x, w, b, act, r)
        if out is None:
            pytest.skip("hipBLASLt route unavailable in this process (callers keep GEMM + tlk_bias_act_nhwc)")
        ref = x.float() @ w.float().t() + b.float()
        if residual:
            ref = ref + r.float()
        ref = F.relu(ref) if act == "relu" else (F.silu(ref) if act == "silu" else ref)
        tol = 2e-3 if dt == torch.float16 else 1.6e-2
        torch.testing.assert_close(out.float(), ref, rtol=tol, atol=tol * 4)
        again = _lib.gemm_bias_act(x, w, b, act, r)                  # cached plan: identical result
        assert torch.equal(out, again)


def test_deepsort_nms_matches_the_reference_vectors_and_oracle(orc):
    """SURVEY 8a G2 (sort/preprocessing.py:6-73, dead code in the reference): kept indices and their order."""
    import torch
    from test_oracle_nms import nms_cases
    from tracklab_amd import _lib
    for boxes, thr, scores, pick in nms_cases():
        got = _lib.deepsort_nms(torch.from_numpy(boxes).cuda(), thr, None if scores is None else torch.from_numpy(scores).cuda())
        assert got.cpu().tolist() == pick
    rng = np.random.default_rng(5)
    for n in (2, 63, 64, 65, 500, 1024):
        boxes = np.concatenate([rng.uniform(0, 900, (n, 2)), rng.uniform(20, 200, (n, 2))], 1)
        scores = rng.permutation(n) / n
        for sc in (scores, None):
            exp = orc.deepsort_nms(boxes, 0.45, sc)
            got = _lib.deepsort_nms(torch.from_numpy(boxes).cuda(), 0.45, None if sc is None else torch.from_numpy(sc).cuda())
            assert got.cpu().tolist() == exp, (n, sc is None)
    assert _lib.deepsort_nms(torch.zeros((0, 4), dtype=torch.float64, device="cuda"), 0.5).numel() == 0
    with pytest.raises(_lib.TlkError):
        _lib.deepsort_nms(torch.zeros((1025, 4), dtype=torch.float64, device="cuda"), 0.5)
