"""ISA audit of the built library: disassembles every gfx950 code object of libmmsa_hip.so and reports, per kernel, the
packed-fp32 instructions (v_pk_fma/mul/add_f32) whose LOW result lane selects the HIGH half of a source register pair
(an `op_sel:[..1..]` operand).  hipcc's SLP vectoriser produces that form from neighbouring scalar FMAs, and it is the form
that returned wrong upper halves in dwpair_gate_kernel under concurrent streams in round 1 (LAB_NOTES.md section 4); the library is
built with -fno-slp-vectorize and must contain none.  python tools/isa_audit.py [path/to/lib.so] -> exit code 1 if any is found."""
import os
import re
import shutil
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = "/opt/rocm/lib/llvm/bin"
PK = re.compile(r"\bv_pk_(fma|mul|add)_f32\b")


def audit(so_path):
    """-> (n_code_objects, n_packed_fp32, [(kernel, instruction text)] of the swizzled ones)"""
    tmp = tempfile.mkdtemp(prefix="mmsa_isa_")
    try:
        so = os.path.join(tmp, "lib.so")
        shutil.copy(so_path, so)
        subprocess.run([os.path.join(LLVM, "llvm-objdump"), "--offloading", so], check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        cos = sorted(f for f in os.listdir(tmp) if "amdgcn" in f)
        bad, npk = [], 0
        for co in cos:
            dis = subprocess.run([os.path.join(LLVM, "llvm-objdump"), "-d", os.path.join(tmp, co)], check=True, capture_output=True, text=True).stdout
            kern = "?"
            for line in dis.splitlines():
                if line.endswith(">:"):
                    kern = line.split("<")[-1][:-2]
                elif PK.search(line):
                    npk += 1
                    if "op_sel:[" in line:
                        bad.append((kern, line.split("//")[0].strip()))
        return len(cos), npk, bad
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


if __name__ == "__main__":
    path = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "multimodal-sam-adapter_amd", "mmsa", "libmmsa_hip.so")
    n, npk, bad = audit(path)
    print(f"{path}: {n} gfx950 code objects, {npk} packed-fp32 instructions, {len(bad)} with a lane-swizzling op_sel")
    for k, ins in bad[:40]:
        print(f"  {k}: {ins}")
    sys.exit(1 if bad else 0)
