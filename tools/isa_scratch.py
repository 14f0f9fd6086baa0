"""Per-kernel register / scratch census of one HIP source compiled for gfx950 (device ISA only):
    python tools/isa_scratch.py multimodal-sam-adapter_amd/csrc/gemm_v2.hip [-DX ...]
prints, per kernel: scratch_ instructions, VGPRs, scratch bytes, s_waitcnt vmcnt(0) count, v_readlane count.  Used by
tests/test_host_cpu.py::test_gemm_kernels_do_not_spill (a scratch reload in front of an LDS-DMA drains the prefetch stream)."""
import os
import re
import subprocess
import sys
import tempfile

FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-Wno-unused-value", "-fno-slp-vectorize", "-w", "-S", "--cuda-device-only"]


def demangle_template(name):
    """_Z14gemm_v2_kernelILb0ELi1ELb1ELi8ELi3EEv10GemmV2Args -> gemm_v2_kernel<0,1,1,8,3> (enough for the kernels of this library)"""
    m = re.match(r"_Z\d+([A-Za-z_0-9]+?)I(.*)Ev", name)
    if not m:
        return name
    args = re.findall(r"L[bi](n?\d+)E", m.group(2))
    return m.group(1) + "<" + ",".join(a.replace("n", "-") for a in args) + ">"


def census(src, extra=()):
    with tempfile.TemporaryDirectory() as tmp:
        out = os.path.join(tmp, "k.s")
        subprocess.run([os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")] + FLAGS + list(extra) + ["-o", out, src], check=True, stderr=subprocess.DEVNULL)
        res, name = {}, None
        for line in open(out):
            m = re.match(r"^(_Z\S+):", line)
            if m:
                name = demangle_template(m.group(1))
                res[name] = dict(scratch=0, vgprs=0, scratch_bytes=0, vmcnt0=0, readlane=0, lines=0)
            if name is None:
                continue
            r = res[name]
            r["lines"] += 1
            if "scratch_" in line and not line.lstrip().startswith((";", ".")):
                r["scratch"] += 1
            if "s_waitcnt vmcnt(0)" in line:
                r["vmcnt0"] += 1
            if "v_readlane_b32" in line:
                r["readlane"] += 1
            m2 = re.search(r"; NumVgprs: (\d+)", line)
            if m2:
                r["vgprs"] = int(m2.group(1))
            m3 = re.search(r"; ScratchSize: (\d+)", line)
            if m3:
                r["scratch_bytes"] = int(m3.group(1))
                name = None
        return res


if __name__ == "__main__":
    for k, v in census(sys.argv[1], sys.argv[2:]).items():
        print(f"{k:44s} scratch_instr {v['scratch']:4d}  vgprs {v['vgprs']:3d}  scratch_bytes {v['scratch_bytes']:4d}  vmcnt(0) {v['vmcnt0']:3d}  readlane {v['readlane']:4d}  lines {v['lines']}")
