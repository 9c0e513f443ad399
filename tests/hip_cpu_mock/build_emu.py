"""TEST INFRASTRUCTURE: builds tests/_build/libmgcmt_emu.so (g++, host only) from the unmodified
kernel sources using the HIP stand-in header in this directory.  See hip/hip_runtime.h."""
import glob
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
OUT_DIR = os.path.join(ROOT, "tests", "_build")
OUT = os.path.join(OUT_DIR, "libmgcmt_emu.so")


def build(force=False):
    """MGCMT_EMU_EXTRA (tuning experiments): extra compiler flags; the library then goes to libmgcmt_emu_extra.so."""
    extra = os.environ.get("MGCMT_EMU_EXTRA", "").split()
    global OUT
    if extra:
        OUT = os.path.join(OUT_DIR, "libmgcmt_emu_extra.so")
        force = True
    srcs = sorted(glob.glob(os.path.join(ROOT, "multigridcmt_amd", "csrc", "*.hip")))
    deps = srcs + glob.glob(os.path.join(ROOT, "multigridcmt_amd", "csrc", "*.h")) + \
        glob.glob(os.path.join(ROOT, "include", "*.h")) + \
        [os.path.join(HERE, "hipmock_runtime.cpp"), os.path.join(HERE, "hip", "hip_runtime.h")]
    if not force and os.path.exists(OUT) and all(os.path.getmtime(OUT) >= os.path.getmtime(d) for d in deps):
        return OUT
    os.makedirs(OUT_DIR, exist_ok=True)
    objs = []
    procs = []
    for src in srcs + [os.path.join(HERE, "hipmock_runtime.cpp")]:
        obj = os.path.join(OUT_DIR, os.path.basename(src) + (".extra.o" if extra else ".o"))
        objs.append(obj)
        cmd = ["g++", "-O2", "-std=c++17", "-fPIC", "-ffp-contract=off", "-x", "c++", "-I", HERE,
               "-I", os.path.join(ROOT, "include"), "-I", os.path.join(ROOT, "multigridcmt_amd", "csrc"),
               "-c", src, "-o", obj] + extra
        procs.append((cmd, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
    for cmd, p in procs:
        out, _ = p.communicate()
        if p.returncode:
            sys.stderr.write(out.decode())
            raise RuntimeError("emulator build failed: " + " ".join(cmd))
    subprocess.check_call(["g++", "-shared", "-o", OUT] + objs + ["-ldl"])
    return OUT


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
