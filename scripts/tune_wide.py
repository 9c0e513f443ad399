"""V-cycle and fine-level pass times with 256-column windows on / off (GPU box)."""
import json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from multigridcmt_amd import _lib
from multigridcmt_amd import plan as planmod
from multigridcmt_amd.operators import laplacian_operator
import glob
libs = [_lib.DEFAULT_LIBRARY] + sorted(glob.glob(os.path.join(ROOT, "variants", "lib_*.so")))
for path in libs:
  _lib.use_library(path)
  print(os.path.basename(path), flush=True)
  for g in (4096, 16384):
      p = planmod.Plan(laplacian_operator(g, "2d") * (-1 / np.pi ** 2), 8, nvec=1)
      p.set_shifts([0.0]); p.fill(0, _lib.SLOT_F, 0, 1.0); p.fill(0, _lib.SLOT_V, 0, 0.0)
      row = {"grid": g}
      for wide in (1, 0, 1, 0):
          p.set_option(_lib.OPT_WIDE, wide)
          p.time_smoother(0, _lib.WJACOBI, 2, 2 / 3, 3)
          row.setdefault("pass2_wide%d_ms" % wide, []).append(round(p.time_smoother(0, _lib.WJACOBI, 2, 2 / 3, 20) / 20, 4))
          for _ in range(4): p.vcycle(2, 2, _lib.WJACOBI, omega=2 / 3, nu_coarse=2)
          p.sync(); t0 = time.perf_counter()
          n = 40 if g <= 4096 else 12
          for _ in range(n): p.vcycle(2, 2, _lib.WJACOBI, omega=2 / 3, nu_coarse=2)
          p.sync(); row.setdefault("cycle_wide%d_ms" % wide, []).append(round((time.perf_counter() - t0) / n * 1e3, 4))
      p.set_option(_lib.OPT_WIDE, 1)
      p.close()
      print(json.dumps(row), flush=True)
