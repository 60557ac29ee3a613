"""Static audit of a BUILT library for the instruction form r03 named as unsafe beside MFMA kernels of another stream (DESIGN.md §2,
profiles/r03_pk_f32_root_cause.md):

    v_pk_mul_f32 / v_pk_add_f32 / v_pk_fma_f32   with   op_sel[src1] = 1      (the LOW result reads the ODD register of the src1 VGPR pair)

On MI355X a single such instruction, isolated by `s_nop 4` on both sides, returns lanes 48-63 of its low result computed with src1 = +0.0 in
~1.5e-4 of its executions while a ResNet-50 forward (MFMA kernels) runs on another stream, and never on an idle GPU
(tools/micro/pk_f32_single.hip).  Every other packed-FP32 form tested (plain, op_sel_hi on src1, op_sel on src0 / src2, neg_lo / neg_hi,
v_pk_mov_b32) was exact in 3.3e8 executions each.

The audit extracts every gfx950 code object of the given shared library / object files (llvm-objdump --offloading), disassembles it and lists the
kernels that contain the form.  Exit status 1 if any is found.  `__graft_entry__.build()` runs it over tracklab_amd/lib/libtlk.so.

    python tools/audit_pk_f32.py tracklab_amd/lib/libtlk.so
"""
import os
import re
import shutil
import subprocess
import sys
import tempfile

def _find_objdump():
    """llvm-objdump of the ROCm toolchain (ROCM_PATH, /opt/rocm, then PATH); None if there is none"""
    for root in (os.environ.get("ROCM_PATH"), "/opt/rocm"):
        if root:
            cand = os.path.join(root, "lib", "llvm", "bin", "llvm-objdump")
            if os.path.exists(cand):
                return cand
    return shutil.which("llvm-objdump")


OBJDUMP = _find_objdump()
PK = re.compile(r"\b(v_pk_(?:mul|add|fma)_f32)\b(.*)")
OPSEL = re.compile(r"op_sel:\[([01,]+)\]")


def unsafe(line):
    """(mnemonic, operands) if `line` is a packed-FP32 arithmetic instruction whose op_sel bit for src1 is set"""
    m = PK.search(line)
    if not m:
        return None
    o = OPSEL.search(m.group(2))
    if not o:
        return None
    bits = o.group(1).split(",")
    if len(bits) > 1 and bits[1] == "1":
        return m.group(1), m.group(2).split("//")[0].strip()
    return None


def code_objects(path, work):
    """paths of the gfx950 code objects bundled in `path` (a host ELF with .hip_fatbin, or a bare code object)"""
    local = os.path.join(work, os.path.basename(path))
    shutil.copy(path, local)
    subprocess.run([OBJDUMP, "--offloading", local], capture_output=True, text=True, check=False)
    found = sorted(os.path.join(work, f) for f in os.listdir(work) if f.startswith(os.path.basename(path) + ".") and "gfx950" in f)
    return found or [local]     # `local` only counts if its disassembly is AMDGCN code (a bare code object), see audit()


def audit(path):
    """{'packed': n, 'unsafe': [(kernel, instruction)], 'objects': n} for every device code object inside `path`"""
    out = {"packed": 0, "unsafe": [], "objects": 0}
    if OBJDUMP is None:
        raise RuntimeError("audit_pk_f32: no llvm-objdump (looked in $ROCM_PATH/lib/llvm/bin, /opt/rocm/lib/llvm/bin and PATH)")
    with tempfile.TemporaryDirectory() as work:
        for co in code_objects(path, work):
            dis = subprocess.run([OBJDUMP, "-d", co], capture_output=True, text=True, check=False).stdout
            if "s_endpgm" not in dis:      # not device code (e.g. the host x86 ELF when nothing was extracted): audits nothing
                continue
            out["objects"] += 1
            kernel = "?"
            for line in dis.split("\n"):
                lab = re.match(r"^[0-9a-f]+ <([^>]+)>:", line)
                if lab:
                    kernel = lab.group(1)
                    continue
                if PK.search(line):
                    out["packed"] += 1
                    hit = unsafe(line)
                    if hit:
                        out["unsafe"].append((kernel, " ".join(hit)))
    return out


def main(argv):
    bad = 0
    for path in argv:
        r = audit(path)
        if r["objects"] == 0:
            print(f"{path}: NO gfx950 code object could be extracted and disassembled - nothing was audited")
            bad += 1
        print(f"{path}: {r['objects']} gfx950 code objects, {r['packed']} packed-FP32 arithmetic instructions, "
              f"{len(r['unsafe'])} with op_sel[src1] = 1")
        per = {}
        for k, ins in r["unsafe"]:
            per.setdefault(k, []).append(ins)
        for k, v in per.items():
            print(f"    {k}: {len(v)}   e.g. {v[0]}")
        bad += len(r["unsafe"])
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1:] or [os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tracklab_amd", "lib", "libtlk.so")]))
