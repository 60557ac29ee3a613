"""-m gpu: CPython's set-iteration order inside the StrongSORT-family association kernels (tlk_pyset.hpp).

`unmatched_tracks = list(set(track_indices) - set(k for k, _ in matches))` (plugins/track/strong_sort/sort/linear_assignment.py:126-127,
plugins/track/bpbreid_strong_sort/sort/linear_assignment.py:127-128) is the row order of the IoU stage; the device emulation must give
the interpreter's order, and the reference-made runs in which that order decides track ids must replay exactly on the GPU."""
import json
import os
import sys

import numpy as np
import pytest

from conftest import GOLDEN

pytestmark = pytest.mark.gpu


def _cases(rng, trials, nmax):
    for trial in range(trials):
        n = int(rng.integers(0, nmax))
        a = np.arange(n, dtype=np.int32) if trial % 2 else np.sort(rng.choice(2 * max(n, 1), size=n, replace=False)).astype(np.int32)
        frac = (0.0, 0.1, 0.5, 0.9, 0.97, 1.0)[trial % 6]
        b = rng.permutation(a)[:int(round(frac * len(a)))].astype(np.int32)
        yield a, b


def test_device_set_order_equals_oracle_and_interpreter(orc):
    import ctypes as C
    from tracklab_amd._lib import pyset_difference_order
    L = orc.lib()
    ip = C.POINTER(C.c_int)
    L.orc_pyset_difference_order.argtypes = [ip, C.c_int, ip, C.c_int, ip]
    rng = np.random.default_rng(5)
    for a, b in _cases(rng, 400, 512):
        out = np.zeros(max(len(a), 1), np.int32)
        m = L.orc_pyset_difference_order(a.ctypes.data_as(ip), len(a), b.ctypes.data_as(ip), len(b), out.ctypes.data_as(ip))
        exp = out[:m].tolist()
        if sys.version_info[:2] == (3, 10):
            assert exp == list(set(a.tolist()) - set(int(x) for x in b.tolist()))
        for force in (False, True):          # shortcut ("no key can wrap -> ascending") and full table emulation
            got = pyset_difference_order(a, b, force_table=force).tolist()
            assert got == exp, (len(a), len(b), force)


def test_plain_strongsort_set_order_golden_on_gpu():
    """The reference run the fuzzer found (32 objects, ids of two tracks born in frame 5 depend on the set order): every row identical."""
    from tracklab_amd._lib import SsortBank
    g = np.load(os.path.join(GOLDEN, "setorder_ssort.npz"))
    D = int(g["dim"])                        # 16 in the fixture; the MFMA cosine tile wants >= 32: zero-pad (norms and dot products unchanged)
    emb = np.zeros((len(g["embeddings"]), 32), np.float32)
    emb[:, :D] = g["embeddings"]
    bank = SsortBank(32, **json.loads(str(g["config"])))
    do, oo = g["det_offsets"], g["out_offsets"]
    for f in range(len(do) - 1):
        r = bank.update(g["dets"][do[f]:do[f + 1]], emb[do[f]:do[f + 1]])
        out = np.column_stack([r["ltrb"], r["track_id"], r["class_id"], r["conf"], r["det_id"]]).astype(np.float64).reshape(-1, 8)
        np.testing.assert_array_equal(out, g["rows"][oo[f]:oo[f + 1]], err_msg=f"frame {f}")
    bank.close()


@pytest.mark.parametrize("seed", [101, 102, 109])
def test_crowded_bpbss_gpu_equals_oracle_where_set_order_matters(orc, seed):
    """135 objects, max_age 300, churn: seeds on which the oracle's ids change from the first frames on if the set order is replaced by
    ascending order. GPU rows must equal the oracle's (CPython order), i.e. the reference's."""
    from tracklab_amd._lib import BpbssBank
    from tracklab_amd.synth import SyntheticStream, ltrb_to_ltwh_rows
    from test_gpu_bpbss import YAML
    K, D = 6, 32
    bank, ref, asc = BpbssBank(K, D, **YAML), orc.StrongSORT(K, D, **YAML), orc.StrongSORT(K, D, **YAML)
    differs = False
    try:
        for fr in SyntheticStream(seed, 135, 40, parts=K, dim=D, with_embeddings=True, miss_prob=0.15, churn_period=3):
            d = fr["dets"]
            args = (d[:, 6].astype(np.int64), ltrb_to_ltwh_rows(d[:, :4]), fr["embeddings"], fr["visibility"], d[:, 4])
            got, exp = bank.update(*args), ref.update(*args)
            orc.python_set_order(False)
            other = asc.update(*args)
            orc.python_set_order(True)
            differs |= len(other) != len(exp) or not np.array_equal(other["track_id"], exp["track_id"])
            assert len(got) == len(exp), fr["frame"]
            for name in ("det_id", "track_id", "hits", "age", "tsu", "state", "matched_name"):
                np.testing.assert_array_equal(got[name], exp[name], err_msg=f"frame {fr['frame']} {name}")
    finally:
        orc.python_set_order(True)
    assert differs, "this seed no longer exercises the set order"
    bank.close()
