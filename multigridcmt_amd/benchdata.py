"""Synthetic right-hand sides of bench.py, defined row block by row block so that a rank of the multi-GPU leg can
produce ITS rows of the very field the one-GPU leg uses without generating the rest (numpy's legacy generator cannot
skip ahead): block b = rows [b*BLOCK_ROWS, (b+1)*BLOCK_ROWS) of the g x g field is RandomState(SEED + b).rand(...)."""
import numpy as np

BLOCK_ROWS = 256
SEED = 1


def rhs_rows(g, row_lo, row_hi):
    """rows [row_lo, row_hi) of the g x g uniform(0,1) field, flattened (k = i*g + j as MGCMTStencilMaker.py:23-24)"""
    row_lo, row_hi = int(row_lo), int(row_hi)
    out = np.empty((row_hi - row_lo) * g)
    pos = 0
    b = row_lo // BLOCK_ROWS
    while b * BLOCK_ROWS < row_hi:
        lo, hi = b * BLOCK_ROWS, min((b + 1) * BLOCK_ROWS, g)
        block = np.random.RandomState(SEED + b).rand((hi - lo) * g)
        a, z = max(lo, row_lo), min(hi, row_hi)
        n = (z - a) * g
        out[pos:pos + n] = block[(a - lo) * g:(z - lo) * g]
        pos += n
        b += 1
    return out


def rhs(g):
    return rhs_rows(g, 0, g)
