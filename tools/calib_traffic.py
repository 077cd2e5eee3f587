#!/usr/bin/env python
"""Known-byte-count kernels for calibrating rocprofv3 FETCH_SIZE / WRITE_SIZE on this box (run under
`rocprofv3 --pmc FETCH_SIZE` resp. `--pmc WRITE_SIZE`): on a 4096^2 level (16.8 M cells)
  rebuild_prob_kernel   reads 4 B/cell (coalesced dwords)            writes 4 B/cell
  rebuild_quad_kernel   reads 4 B/cell + neighbours (cache hits)      writes 16 B/cell
  pack_cells_kernel     reads 8 B/cell                                writes 8 B/cell
  fill_level_kernel     reads nothing                                 writes 28 B/cell
"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from hector_slam_amd import capi
S = 4096
m = capi.MapRepMultiMap(0.05, S, S, 1)
lo = np.random.default_rng(0).normal(0, 2, (S, S)).astype(np.float32)
for _ in range(3):
    m.upload_level(0, lo)  # -> rebuild_prob_kernel + rebuild_quad_kernel
buf = np.empty((S, S, 2), np.float32)
for _ in range(3):
    capi._check(m._lib.hsm_download_cells(m._h, 0, 0, 0, S - 1, S - 1, buf.ctypes.data, S), "download_cells")
m.reset()
print("cells", S * S)
