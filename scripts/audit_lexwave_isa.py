"""ISA audit of the lexicographic pipelines (kernels_lexwave.hip, kernels_lexband.hip).

The row loop's loads are asm statements the compiler does not count: nothing waits for them but the kernel's own
s_waitcnt.  That is only sound while every destination register of such a load stays the slot's register for the whole
loop.  If the compiler ever routes a destination through a temporary (a phi between two alternative load sequences, a
spill, a re-materialised copy), the temporary is read while its load is in flight and is then reused — as an address
in the worst case, which is a memory fault on the GPU.  This script compiles the kernel for gfx950 and checks, for the
row loop (from its header on), that no destination of a hand-counted load is
  - the source of a plain v_mov_b32 (a copy),
  - the address operand of any global load / store,
  - stored to scratch,
  - shared by two loads of the loop,
and that the compiler put no partial vmcnt wait of its own into the loop.
Exit status 0 = clean.  Usage: python scripts/audit_lexwave_isa.py [file.s [kernel name]]"""
import os, re, subprocess, sys, tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "multigridcmt_amd", "csrc")


KERNELS = {"k_lex_wave": "kernels_lexwave.hip", "k_lex_band": "kernels_lexband.hip"}


def compile_isa(out, source="kernels_lexwave.hip"):
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    cmd = [hipcc, "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-I" + os.path.join(ROOT, "include"), "-I" + CSRC,
           "-ffp-contract=off", "-S", "--cuda-device-only", "-o", out, os.path.join(CSRC, source)]
    subprocess.run(cmd, check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)


def audit_loop(loop):
    dests = set()
    seen = {}
    for n, t in enumerate(loop):
        # (hand-counted = inside an asm statement: the compiler's own loads can take the same scalar-base form)
        m = re.search(r"global_load_dwordx2 v\[(\d+):(\d+)\], v\d+, s\[", t) if n and "#ASMSTART" in loop[n - 1] else None
        if m:
            dests.update({int(m.group(1)), int(m.group(2))})
            seen[m.group(1)] = seen.get(m.group(1), 0) + 1
    bad = []
    # every slot has registers of its own: a destination that two loads of the loop share is a temporary the slot is
    # copied out of afterwards — while the load is in flight
    for reg, count in seen.items():
        if count > 1:
            bad.append("v[%s:..] is the destination of %d hand-counted loads" % (reg, count))
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
    # the compiler's own waits inside the loop must be full drains of the slow path only: a partial one (vmcnt(N), N > 1)
    # means it believes a load pending on a slot register and throttles the pipeline on the hardware's counter
    for i, t in enumerate(loop):
        m = re.search(r"s_waitcnt vmcnt\((\d+)\)", t)
        if m and int(m.group(1)) > 1 and "#ASMSTART" not in loop[i - 1]:
            bad.append("compiler-inserted " + t)
    return dests, bad


def audit(path, kernel="k_lex_wave"):
    lines = open(path).read().split("\n")
    starts = [i for i, l in enumerate(lines) if re.match(r"^_ZN5mgcmt.*" + kernel + r".*:", l)]
    report = []
    for k, s in enumerate(starts):
        body = lines[s:(starts[k + 1] if k + 1 < len(starts) else len(lines))]
        # every top-level loop with hand-counted waits is audited by itself (kernels_lexpair.hip has one per wave role: the
        # roles are different waves, the same register numbers in both mean nothing)
        headers = [i for i, l in enumerate(body) if re.match(r"^\.LBB\d+_\d+:", l) and "Loop Header: Depth=1" in l]
        segments = []
        for h, start in enumerate(headers):
            seg = body[start:(headers[h + 1] if h + 1 < len(headers) else len(body))]
            if any((m := re.search(r"s_waitcnt vmcnt\((\d+)\)", l)) and int(m.group(1)) >= 6 and n and "#ASMSTART" in seg[n - 1] for n, l in enumerate(seg)):
                segments.append(seg)
        if not segments:
            report.append((k, 0, ["no loop with hand-counted waits found"]))
            continue
        dests, bad = set(), []
        for seg in segments:
            d, b = audit_loop([l.strip() for l in seg])
            dests |= d
            bad += b
        report.append((k, len(dests), bad))
    return report


def main():
    reports = []
    if len(sys.argv) > 1:
        kernel = sys.argv[2] if len(sys.argv) > 2 else "k_lex_wave"
        reports.append((kernel, audit(sys.argv[1], kernel)))
    else:
        with tempfile.TemporaryDirectory() as d:
            for kernel, source in KERNELS.items():
                path = os.path.join(d, kernel + ".s")
                compile_isa(path, source)
                reports.append((kernel, audit(path, kernel)))
    rc = 0
    for kernel, rep in reports:
        if len(rep) < 3:
            print("expected three instantiations of %s, found %d" % (kernel, len(rep)))
            rc = 1
        for k, nd, bad in rep:
            print("%s instantiation %d: %d destination registers of hand-counted loads, %d suspicious uses" % (kernel, k, nd, len(bad)))
            for t in bad[:8]:
                print("    " + t)
            if bad or nd == 0:
                rc = 1
    return rc


if __name__ == "__main__":
    sys.exit(main())
