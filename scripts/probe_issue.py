"""What ONE wave per SIMD can issue (kinds 20..23 of mgcmt_bandwidth_probe): nanoseconds per instruction for a chain of
dependent double FMAs, eight independent chains, dependent 32-bit vector adds, dependent scalar adds — with one wave on
the chip, one per CU, and four per CU (one per SIMD)."""
import json, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from multigridcmt_amd.operators import laplacian_operator
from multigridcmt_amd.plan import Plan
p = Plan(laplacian_operator(256, "2d"), 8, nvec=1)
names = {20: "dependent f64 fma", 21: "8 independent f64 fma chains", 22: "dependent v_add_u32", 23: "dependent s_add_u32"}
for kind in (20, 21, 22, 23):
    row = {"probe": names[kind]}
    for blocks in (1, 256, 1024, 4096):
        ms = p.bandwidth_probe(0, kind, blocks, 3)
        row["ns_per_instruction_%d_waves" % blocks] = round(ms * 1e6 / (64 * 20000), 3)
    print(json.dumps(row), flush=True)
p.close()
