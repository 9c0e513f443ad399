import json, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from multigridcmt_amd import _lib
from multigridcmt_amd import plan as planmod
from multigridcmt_amd.operators import laplacian_operator
g = 16384
p = planmod.Plan(laplacian_operator(g, "2d") * (-1 / np.pi ** 2), 8, nvec=1)
p.set_shifts([0.0]); p.fill(0, _lib.SLOT_V, 0, 1.0); p.fill(0, _lib.SLOT_F, 0, 2.0)
for rows in (328, 656):
    row = {"rows": rows}
    for kind, name, bpp in ((3, "march_read1", 8), (4, "march_read2", 16), (5, "march_triad", 24),
                            (6, "ovl_read1", 8), (7, "ovl_read2", 16), (8, "ovl_triad", 24),
                            (9, "w112_read1", 8), (10, "w112_read2", 16), (11, "w112_triad", 24),
                            (12, "w96_read1", 8), (13, "w96_read2", 16), (14, "w96_triad", 24),
                            (15, "w120_read1", 8), (16, "w120_read2", 16), (17, "w120_triad", 24)):
        ms = p.bandwidth_probe(0, kind, rows, 10)
        row[name + "_TBs"] = round(g * g * bpp / (ms * 1e-3) / 1e12, 3)
    print(json.dumps(row), flush=True)
for blocks in (1024, 4096):
    row = {"linear_blocks": blocks}
    for kind, name, bpp in ((0, "copy", 16), (1, "triad", 24), (2, "read", 8)):
        ms = p.bandwidth_probe(0, kind, blocks, 10)
        row[name + "_TBs"] = round(g * g * bpp / (ms * 1e-3) / 1e12, 3)
    print(json.dumps(row), flush=True)
