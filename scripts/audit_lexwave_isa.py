"""ISA audit of the lexicographic wave pipeline (kernels_lexwave.hip).

The row loop's loads are asm statements the compiler does not count: nothing waits for them but the kernel's own
s_waitcnt.  That is only sound while every destination register of such a load stays the slot's register for the whole
loop.  If the compiler ever routes a destination through a temporary (a phi between two alternative load sequences, a
spill, a re-materialised copy), the temporary is read while its load is in flight and is then reused — as an address
in the worst case, which is a memory fault on the GPU.  This script compiles the kernel for gfx950 and checks, for the
region after the prologue's last vmcnt(0), that no destination of a hand-counted load is
  - the source of a plain v_mov_b32 (a copy),
  - the address operand of any global load / store,
  - stored to scratch.
Exit status 0 = clean.  Usage: python scripts/audit_lexwave_isa.py [file.s]"""
import os, re, subprocess, sys, tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "multigridcmt_amd", "csrc")


def compile_isa(out):
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    cmd = [hipcc, "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-I" + os.path.join(ROOT, "include"), "-I" + CSRC,
           "-ffp-contract=off", "-S", "--cuda-device-only", "-o", out, os.path.join(CSRC, "kernels_lexwave.hip")]
    subprocess.run(cmd, check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)


def audit(path):
    lines = open(path).read().split("\n")
    starts = [i for i, l in enumerate(lines) if re.match(r"^_ZN5mgcmt.*k_lex_wave.*:", l)]
    report = []
    for k, s in enumerate(starts):
        body = lines[s:(starts[k + 1] if k + 1 < len(starts) else len(lines))]
        counted = [i for i, l in enumerate(body) if (m := re.search(r"s_waitcnt vmcnt\((\d+)\)", l)) and int(m.group(1)) >= 20]
        if not counted:
            report.append((k, 0, ["no hand-counted wait found"]))
            continue
        prologue_end = [i for i, l in enumerate(body[:counted[0]]) if "s_waitcnt vmcnt(0)" in l][-1]
        loop = [l.strip() for l in body[prologue_end:]]
        dests = set()
        for t in loop:
            m = re.search(r"global_load_dwordx2 v\[(\d+):(\d+)\], v\d+, s\[", t)
            if m:
                dests.update({int(m.group(1)), int(m.group(2))})
        bad = []
        for t in loop:
            m = re.match(r"v_mov_b32\S*\s+v(\d+), v(\d+)", t)
            if m and int(m.group(2)) in dests and "dpp" not in t:
                bad.append(t)
            m = re.search(r"global_store_dwordx2 v\[(\d+):(\d+)\]", t)
            if m and int(m.group(1)) in dests:
                bad.append(t)
            m = re.search(r"global_load_dwordx2 v\[\d+:\d+\], v\[(\d+):(\d+)\], off", t)
            if m and int(m.group(1)) in dests:
                bad.append(t)
            m = re.search(r"global_load_dwordx2 v\[\d+:\d+\], v(\d+), s\[", t)
            if m and int(m.group(1)) in dests:
                bad.append(t)
            m = re.search(r"scratch_store\S* .*?v\[?(\d+)", t)
            if m and int(m.group(1)) in dests:
                bad.append(t)
        report.append((k, len(dests), bad))
    return report


def main():
    if len(sys.argv) > 1:
        path = sys.argv[1]
        rep = audit(path)
    else:
        with tempfile.TemporaryDirectory() as d:
            path = os.path.join(d, "lexwave.s")
            compile_isa(path)
            rep = audit(path)
    rc = 0
    if len(rep) < 3:
        print("expected three instantiations of k_lex_wave, found", len(rep))
        rc = 1
    for k, nd, bad in rep:
        print("k_lex_wave instantiation %d: %d destination registers of hand-counted loads, %d suspicious uses" % (k, nd, len(bad)))
        for t in bad[:8]:
            print("    " + t)
        if bad or nd == 0:
            rc = 1
    return rc


if __name__ == "__main__":
    sys.exit(main())
