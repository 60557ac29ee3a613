"""Driver of tools/micro/pk_f32_single.hip (r03): one packed-FP32 instruction form per kernel, alone on the GPU and under a bf16 ResNet-50 forward
on another stream; counts the passes with any wrong lane and dumps the logged events.
Run on the GPU box:  python tools/micro/pk_f32_single.py [out.json] [launches_under_load]"""
import ctypes as C, json, os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from tracklab_amd.backbones.reid import part_based_reid

HERE = os.path.dirname(os.path.abspath(__file__))
L = C.CDLL(os.path.join(HERE, "libpk_f32_single.so"))
L.single_launch.argtypes = [C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
OPS = {
    0: "v_pk_mul_f32 d, a, b",
    1: "v_pk_mul_f32 d, a, b op_sel:[0,1] op_sel_hi:[1,0]   (src1 halves swapped)",
    2: "v_pk_mul_f32 d, a, b op_sel:[0,1]                    (src1 hi for both)",
    3: "v_pk_mul_f32 d, a, b op_sel_hi:[1,0]                 (src1 lo for both)",
    4: "v_pk_mul_f32 d, a, b op_sel:[1,0] op_sel_hi:[0,1]   (src0 halves swapped)",
    5: "v_pk_fma_f32 d, a, b, c neg_lo:[0,0,1] neg_hi:[0,0,1]",
    6: "v_pk_fma_f32 d, a, b, c",
    7: "v_pk_add_f32 d, a, b op_sel:[0,1] op_sel_hi:[1,0]",
    8: "v_pk_fma_f32 d, a, b, c op_sel:[0,0,1] op_sel_hi:[1,1,0] (src2 halves swapped)",
    9: "pair: v_pk_mul_f32 d, a, b (src1 swapped); s_nop 0; v_pk_fma_f32 e, c, b, d neg_lo/neg_hi on src2",
    10: "pair: v_pk_mul_f32 t, b, c; s_nop 0; v_pk_mul_f32 d, a, t (src1 swapped)",
    11: "v_pk_mov_b32 d, b, b op_sel:[1,0]",
    12: "v_pk_mul_f32 d, a, b neg_lo:[0,1] neg_hi:[0,1]",
    13: "v_pk_fma_f32 d, a, b, c neg_lo:[0,0,1]",
    14: "control: scalar twins of 9",
    15: "control: 9 with plain packed forms on pre-swizzled / pre-negated registers",
}
EV = L.single_event_dwords()
CAP = 128
out_path = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/pk_f32_single.json"
n_load = int(sys.argv[2]) if len(sys.argv) > 2 else 8
reid = part_based_reid(1, 128, device="cuda", dtype=torch.bfloat16, channels_last=True)
crops = torch.randn(96, 3, 256, 128, device="cuda", dtype=torch.bfloat16).to(memory_format=torch.channels_last)
with torch.no_grad():
    reid(crops)
torch.cuda.synchronize()
side = torch.cuda.Stream()
result = {"ops": {}, "device": torch.cuda.get_device_name(0)}
for op in range(L.single_n_ops()):
    row = {"what": OPS.get(op, "?")}
    for mode, reps in (("alone", 2), ("load", n_load)):
        cnt = torch.zeros(4, dtype=torch.int32, device="cuda")
        log = torch.zeros(CAP * EV, dtype=torch.int32, device="cuda")
        for rep in range(reps):
            if mode == "load":
                with torch.no_grad():
                    for _ in range(2):
                        reid(crops)
            rc = L.single_launch(op, 512, 20000, cnt.data_ptr(), log.data_ptr(), CAP, C.c_void_p(side.cuda_stream))
            assert rc == 0, rc
            torch.cuda.synchronize()
        n, bad, logged = cnt.tolist()[:3]
        row[mode] = {"passes": n, "bad": bad}
        if bad:
            ev = log.cpu().numpy().view(np.uint32).reshape(CAP, EV)[:min(logged, CAP)]
            row[mode]["events"] = ev.tolist()
    result["ops"][op] = row
    print(f"op {op:2d} {row['what']}: alone {row['alone']['bad']} of {row['alone']['passes']} passes wrong; "
          f"under the ResNet-50 forward {row['load']['bad']} of {row['load']['passes']}", flush=True)
os.makedirs(os.path.dirname(out_path) or ".", exist_ok=True)
with open(out_path, "w") as f:
    json.dump(result, f)
print("events ->", out_path)
