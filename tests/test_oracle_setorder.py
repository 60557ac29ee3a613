"""CPU: the CPython set-iteration order behind the StrongSORT-family cascades (oracle/src/pyset.c).
`unmatched_tracks = list(set(track_indices) - set(k for k, _ in matches))` (strong_sort/sort/linear_assignment.py:126, same line in
bpbreid_strong_sort) is ascending only while every track index is below the result set's table size; its order is the row order of the
IoU stage and decides which of two tracks born in one frame gets the lower id."""
import ctypes as C
import json
import os
import sys

import numpy as np
import pytest

from conftest import GOLDEN


def test_pyset_emulation_equals_this_interpreters_sets(orc):
    if sys.version_info[:2] != (3, 10):
        pytest.skip("pyset.c restates CPython 3.10's setobject.c (the goldens were made under 3.10.12)")
    L = orc.lib()
    ip = C.POINTER(C.c_int)
    L.orc_pyset_difference_order.argtypes = [ip, C.c_int, ip, C.c_int, ip]
    rng = np.random.default_rng(1)
    for trial in range(3000):
        n = int(rng.integers(0, 200))
        a = np.arange(n, dtype=np.int32) if trial % 2 else np.sort(rng.choice(2 * max(n, 1), size=n, replace=False)).astype(np.int32)
        b = rng.permutation(a)[:int(rng.integers(0, len(a) + 1))].astype(np.int32)
        out = np.zeros(max(len(a), 1), np.int32)
        m = L.orc_pyset_difference_order(a.ctypes.data_as(ip), len(a), b.ctypes.data_as(ip), len(b), out.ctypes.data_as(ip))
        assert out[:m].tolist() == list(set(a.tolist()) - set(int(x) for x in b.tolist())), trial


def _replay(orc, g):
    trk = orc.PlainStrongSORT(int(g["dim"]), **json.loads(str(g["config"])))
    do, oo = g["det_offsets"], g["out_offsets"]
    bad = []
    for f in range(len(do) - 1):
        out = trk.update(g["dets"][do[f]:do[f + 1]], g["embeddings"][do[f]:do[f + 1]])
        exp = g["rows"][oo[f]:oo[f + 1]]
        if out.shape != exp.shape or not np.array_equal(out, exp):
            bad.append(f)
    return bad


def test_plain_strongsort_set_order_case(orc):
    """With CPython's order (the oracle's default, and what tlk_ssort / tlk_bpbss implement on the device) the oracle reproduces the
    reference on the run the fuzzer found; with ascending order the two tracks born in frame 5 swap ids and keep them swapped."""
    g = np.load(os.path.join(GOLDEN, "setorder_ssort.npz"))
    try:
        assert orc.lib().orc_get_python_set_order() == 1
        assert _replay(orc, g) == []
        orc.python_set_order(False)
        bad = _replay(orc, g)
        assert bad and bad[0] == 5
    finally:
        orc.python_set_order(True)
