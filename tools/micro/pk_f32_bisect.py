"""Driver of tools/micro/pk_f32_bisect.hip (r03): every variant of the LK dx/dy instruction sequence alone on the GPU and under a bf16 ResNet-50
forward on another stream; counts the passes whose result differs from the scalar twin in any lane and dumps the logged events.
Run on the GPU box:  python tools/micro/pk_f32_bisect.py [out.json] [launches_under_load] [variants, comma separated]"""
import ctypes as C, json, os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from tracklab_amd.backbones.reid import part_based_reid

HERE = os.path.dirname(os.path.abspath(__file__))
L = C.CDLL(os.path.join(HERE, "libpk_f32_bisect.so"))
L.bisect_launch.argtypes = [C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
VARIANTS = {
    0: "the compiler's sequence",
    1: "s_nop 1 after each v_cvt_f32_f64",
    2: "s_nop 1 after the conversions and between all packed instructions",
    3: "conversions replaced by v_mov of precomputed floats (no f64 producer)",
    4: "plain packed forms only (no op_sel / neg / SGPR-pair operand)",
    5: "conversions kept, scalar twins instead of packed instructions",
    6: "s_nop 1 only after the conversion ADJACENT to the first packed instruction (v118)",
    7: "s_nop 1 only after the first conversion (v119, 5 instructions ahead of its consumer)",
    8: "first conversion's destination moved out of its own source pair",
    9: "reproducer hygiene: every readable register initialised + s_waitcnt before the sequence",
    10: "SGPR-pair operands replaced by VGPR pairs (op_sel swap + neg kept)",
    11: "half-swapping op_sel and neg replaced (SGPR-pair op_sel_hi kept)",
    12: "the compiler's own s_nop 0 removed",
    13: "two v_nop instead of s_nop 1 after the adjacent conversion",
    14: "A: half-swapping op_sel kept, neg_lo/neg_hi replaced by v_xor",
    15: "B: neg_lo/neg_hi kept, half-swapping op_sel replaced by a pre-swizzled pair",
}
EV = L.bisect_event_dwords()
CAP = 256
out_path = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/pk_f32_bisect.json"
n_load = int(sys.argv[2]) if len(sys.argv) > 2 else 8
only = [int(v) for v in sys.argv[3].split(",")] if len(sys.argv) > 3 else None
reid = part_based_reid(1, 128, device="cuda", dtype=torch.bfloat16, channels_last=True)
crops = torch.randn(96, 3, 256, 128, device="cuda", dtype=torch.bfloat16).to(memory_format=torch.channels_last)
with torch.no_grad():
    reid(crops)
torch.cuda.synchronize()
side = torch.cuda.Stream()
result = {"variants": {}, "device": torch.cuda.get_device_name(0)}
for var in range(L.bisect_n_variants()):
    if only is not None and var not in only:
        continue
    row = {"what": VARIANTS.get(var, "?")}
    for mode, reps in (("alone", 3), ("load", n_load)):
        cnt = torch.zeros(4, dtype=torch.int32, device="cuda")
        log = torch.zeros(CAP * EV, dtype=torch.int32, device="cuda")
        for rep in range(reps):
            if mode == "load":
                with torch.no_grad():
                    for _ in range(2):
                        reid(crops)
            rc = L.bisect_launch(var, 512, 20000, cnt.data_ptr(), log.data_ptr(), CAP, C.c_void_p(side.cuda_stream))
            assert rc == 0, rc
            torch.cuda.synchronize()
        n, bad, logged = cnt.tolist()[:3]
        row[mode] = {"passes": n, "bad": bad}
        if bad:
            ev = log.cpu().numpy().view(np.uint32).reshape(CAP, EV)[:min(logged, CAP)]
            row[mode]["events"] = ev.tolist()
    result["variants"][var] = row
    print(f"variant {var:2d} ({row['what']}): alone {row['alone']['bad']} of {row['alone']['passes']} passes wrong; "
          f"under the ResNet-50 forward {row['load']['bad']} of {row['load']['passes']}", flush=True)
os.makedirs(os.path.dirname(out_path) or ".", exist_ok=True)
with open(out_path, "w") as f:
    json.dump(result, f)
print("events ->", out_path)
