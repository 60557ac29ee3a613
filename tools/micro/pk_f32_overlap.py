"""Driver of tools/micro/pk_f32_overlap.hip: the packed-FP32 recurrence alone on the GPU, then under a ResNet-50 forward (bf16 convolutions:
MFMA kernels) running on another stream.  Run on the GPU box:  python tools/micro/pk_f32_overlap.py"""
import ctypes as C, os, sys
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from tracklab_amd.backbones.reid import part_based_reid

L = C.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "libpk_f32_overlap.so"))
L.pk_launch.argtypes = [C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
L2 = C.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "libpk_f32_lk_sequence.so"))
L2.lkseq_launch.argtypes = [C.c_int, C.c_int, C.c_void_p, C.c_void_p]
L2.lkseq_launch_var.argtypes = [C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
VARIANTS = {0: "the compiler's sequence", 1: "s_nop 1 after each v_cvt_f32_f64", 2: "s_nop 1 after the conversions and between all packed instructions",
            3: "conversions replaced by v_mov of precomputed floats", 4: "plain packed forms on pre-swizzled registers (no op_sel / neg)",
            5: "conversions kept, scalar twins instead of packed instructions"}
reid = part_based_reid(1, 128, device="cuda", dtype=torch.bfloat16, channels_last=True)
crops = torch.randn(96, 3, 256, 128, device="cuda", dtype=torch.bfloat16).to(memory_format=torch.channels_last)
with torch.no_grad():
    reid(crops)
torch.cuda.synchronize()
side = torch.cuda.Stream()
for mode in ("alone", "under a ResNet-50 forward on another stream", "alone"):
    cnt = torch.zeros(4, dtype=torch.int32, device="cuda")
    for rep in range(8):
        if mode != "alone":
            with torch.no_grad():
                for _ in range(2):
                    reid(crops)
        L.pk_launch(512, 200, 500, cnt.data_ptr(), C.c_void_p(side.cuda_stream))
        torch.cuda.synchronize()
    n, bad_pk, bad_sc, bad_cmp = cnt.tolist()
    print(f"{mode:46s}: passes {n}  packed result not identical in all lanes {bad_pk}  scalar result not identical in all lanes {bad_sc}  "
          f"packed != scalar {bad_cmp}", flush=True)

for mode in ("alone", "under a ResNet-50 forward on another stream", "alone"):
    cnt = torch.zeros(4, dtype=torch.int32, device="cuda")
    for rep in range(8):
        if mode != "alone":
            with torch.no_grad():
                for _ in range(2):
                    reid(crops)
        L2.lkseq_launch(512, 20000, cnt.data_ptr(), C.c_void_p(side.cuda_stream))
        torch.cuda.synchronize()
    n, bad = cnt.tolist()[:2]
    print(f"LK dx/dy sequence, {mode:46s}: passes {n}  result not identical in all lanes {bad}", flush=True)

if "--bisect" in sys.argv:          # which ingredient of the sequence fails under load?  (about 25 s of GPU time)
    for var, what in VARIANTS.items():
        res = []
        for mode in ("alone", "load"):
            cnt = torch.zeros(4, dtype=torch.int32, device="cuda")
            for rep in range(6):
                if mode == "load":
                    with torch.no_grad():
                        for _ in range(2):
                            reid(crops)
                L2.lkseq_launch_var(var, 512, 20000, cnt.data_ptr(), C.c_void_p(side.cuda_stream))
                torch.cuda.synchronize()
            res.append(cnt.tolist()[:2])
        print(f"variant {var} ({what}): alone {res[0][1]} of {res[0][0]} passes lane-dependent; under the ResNet-50 forward {res[1][1]} of {res[1][0]}", flush=True)
