"""Static check of one kernel's ISA (hipcc -S output): does any instruction touch a VGPR that a still-outstanding vector-memory load is going to
write?  gfx9 retires vector-memory operations in order, so after `s_waitcnt vmcnt(N)` only the N youngest are outstanding.  The walk is linear
(branches ignored: skipping code only removes loads, so linear order is the conservative one for forward branches; a backward branch keeps whatever is
outstanding at the bottom of the loop when it re-enters the top, which the walk models by running the loop body twice).
python tools/isa_waitcnt_check.py file.s kernel_name_substring"""
import re
import sys


def regs(tok):
    """VGPR numbers named by an operand token like v12 or v[10:13]."""
    m = re.fullmatch(r"v(\d+)", tok)
    if m:
        return {int(m.group(1))}
    m = re.fullmatch(r"v\[(\d+):(\d+)\]", tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    return set()


def kernel_lines(path, name):
    out, on = [], False
    for ln in open(path):
        if re.match(r"^_Z\w*:", ln) or re.match(r"^\w+:\s*;\s*@", ln):
            on = name in ln
        if on:
            out.append(ln.rstrip("\n"))
            if "s_endpgm" in ln:
                break
    return out


def check(lines, passes=2):
    pending = []   # FIFO of (line_no, text, dest regs) -- stores have no dest
    findings = []
    for _ in range(passes):
        for no, ln in enumerate(lines):
            t = ln.strip()
            if not t or t.startswith(";") or t.startswith(".") or t.endswith(":"):
                continue
            op = t.split()[0]
            toks = [x.strip(",") for x in t.split()[1:]]
            if op == "s_waitcnt":
                m = re.search(r"vmcnt\((\d+)\)", t)
                if m:
                    n = int(m.group(1))
                    pending = pending[len(pending) - n:] if n else []
                continue
            touched = set()
            for x in toks:
                touched |= regs(x)
            for (pno, ptxt, dst) in pending:
                if dst & touched:
                    findings.append((no, t, pno, ptxt, sorted(dst & touched)))
            if op.startswith(("global_load", "buffer_load", "flat_load", "scratch_load")):
                pending.append((no, t, regs(toks[0]) if toks else set()))
            elif op.startswith(("global_store", "buffer_store", "flat_store", "scratch_store", "global_atomic", "buffer_atomic")):
                pending.append((no, t, set()))
    return findings


if __name__ == "__main__":
    ls = kernel_lines(sys.argv[1], sys.argv[2])
    f = check(ls)
    seen = set()
    print(f"{sys.argv[2]}: {len(ls)} lines, {sum(1 for x in ls if 'v_pk_' in x)} packed fp32 instructions")
    for (no, t, pno, ptxt, rr) in f:
        if (no, pno) in seen:
            continue
        seen.add((no, pno))
        print(f"  line {no}: {t}\n      touches v{rr} while outstanding: line {pno}: {ptxt}")
    print(f"  {len(seen)} finding(s)")
