"""Static audit: packed-FP32 instructions (v_pk_mul/add/fma_f32) that consume, within two instructions and without an s_nop in between, a VGPR
written by a double-precision-pipeline or transcendental instruction -- the producer -> consumer pattern of profiles/r02_pk_f32_overlap.md.
Compiles each source with the DEFAULT flags (packed FP32 allowed) and scans the gfx950 ISA.  python tools/audit_pk_f32.py tracklab_amd/csrc/*.hip"""
import re, subprocess, sys
def regs(tok):
    out=set()
    for m in re.finditer(r'v\[(\d+):(\d+)\]|v(\d+)', tok):
        if m.group(1): out |= set(range(int(m.group(1)), int(m.group(2))+1))
        else: out.add(int(m.group(3)))
    return out
for f in sys.argv[1:]:
    asm = subprocess.run(["/opt/rocm/bin/hipcc","--offload-arch=gfx950","-O3","-std=c++17","-ffp-contract=off","-I/root/repo/include","-I/root/repo/tracklab_amd/csrc","-S","--cuda-device-only","-o","-",f],capture_output=True,text=True).stdout
    lines=[l.strip() for l in asm.split("\n")]
    ins=[l for l in lines if l and not l.startswith((";",".","//")) and not l.endswith(":")]
    n_pk=0; hits=[]
    for i,l in enumerate(ins):
        if re.match(r'v_pk_(mul|add|fma)_f32', l):
            n_pk+=1
            ops=l.split(None,1)[1].split(",")
            src=set()
            for o in ops[1:]: src|=regs(o)
            # look back up to 2 real instructions (s_nop counts as separation)
            for back in (1,2):
                if i-back<0: break
                p=ins[i-back]
                if p.startswith("s_nop"): break
                pm=p.split(None,1)
                if len(pm)<2: continue
                dst=regs(pm[1].split(",")[0])
                if dst & src and ("_f64" in pm[0] or pm[0].startswith(("v_rcp","v_sqrt","v_rsq","v_exp","v_log","v_cvt_f32_f64"))):
                    hits.append((back,p,l)); break
    print(f, "packed f32 ops:", n_pk, "| fed by a DP/trans result 1-2 instructions earlier:", len(hits))
    for h in hits[:6]: print("    ", h)
