"""Build libmmsa_hip.so in-tree with hipcc for gfx950 (cross-compiles without a GPU)."""
import glob
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
INCLUDE = os.path.join(os.path.dirname(HERE), "include")   # the C ABI's header: csrc/common.h includes it, every definition is compiled against its prototype
OUT = os.path.join(HERE, "mmsa", "libmmsa_hip.so")


# EVERY source is compiled with -fno-slp-vectorize.  hipcc's SLP vectoriser packs neighbouring scalar fp32 FMAs into
# v_pk_fma_f32 / v_pk_mul_f32 whose LOW lane reads the HIGH half of a register pair (op_sel swizzles); dwpair_gate_kernel built
# that way returned wrong upper halves under concurrent streams in round 1 (header of csrc/conv_pair.hip), root cause unknown, and
# the same instruction form was present in conv.hip / tail.hip / head.hip / msda.hip / segment.hip (tools/isa_audit.py: 263 of them).
# Built without the vectoriser the library contains none; tests/test_host_cpu.py::test_no_lane_swizzled_packed_fp32 keeps it so.
# The explicit, lane-wise packed arithmetic of common.h (gelu2, split2: op_sel_hi broadcasts only) is a different instruction
# form and has never differed in the concurrency stress runs.
CXXFLAGS = ["-O3", "-fPIC", "-std=c++17", "-Wno-unused-value", "-fno-slp-vectorize"]


def source_digest():
    """sha1 over the kernel sources (csrc/*.hip, *.h, *.inc: names and bytes).  The counter profiles under profiles/ carry the digest of the
    sources they were measured on; bench.py attaches a profile to its JSON line only when it equals the digest of the sources in the tree."""
    import hashlib
    h = hashlib.sha1()
    for f in sorted(glob.glob(os.path.join(CSRC, "*.hip")) + glob.glob(os.path.join(CSRC, "*.h")) + glob.glob(os.path.join(CSRC, "*.inc"))) + sorted(glob.glob(os.path.join(INCLUDE, "*.h"))):
        h.update(os.path.basename(f).encode())
        h.update(open(f, "rb").read())
    return h.hexdigest()


def needs_build():
    if not os.path.exists(OUT):
        return True
    t = os.path.getmtime(OUT)
    srcs = glob.glob(os.path.join(CSRC, "*.hip")) + glob.glob(os.path.join(CSRC, "*.h")) + glob.glob(os.path.join(CSRC, "*.inc")) + glob.glob(os.path.join(INCLUDE, "*.h"))
    return any(os.path.getmtime(s) > t for s in srcs)


def build(force=False, verbose=True):
    if not force and not needs_build():
        return OUT
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    srcs = sorted(glob.glob(os.path.join(CSRC, "*.hip")))
    objs = []
    procs = []
    os.makedirs(os.path.join(HERE, "build"), exist_ok=True)
    for s in srcs:
        o = os.path.join(HERE, "build", os.path.basename(s).replace(".hip", ".o"))
        objs.append(o)
        cmd = [hipcc, "--offload-arch=gfx950"] + CXXFLAGS + ["-c", s, "-o", o]
        procs.append((s, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
    for s, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            sys.stderr.write(out.decode())
            raise RuntimeError(f"hipcc failed on {s}")
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", OUT] + objs
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
    if r.returncode != 0:
        sys.stderr.write(r.stdout.decode())
        raise RuntimeError("hipcc link failed")
    if verbose:
        print(f"built {OUT}")
    return OUT


if __name__ == "__main__":
    build(force="--force" in sys.argv)
