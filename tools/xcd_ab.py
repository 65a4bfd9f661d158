#!/usr/bin/env python3
"""A/B of the XCD-aware visiting order (NMI_OPT_XCD_TILING) in one process: kernel time, interleaved rounds."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import orbslam2_nmi_amd as nmi
from orbslam2_nmi_amd import synthetic as sy
mode = sys.argv[1] if len(sys.argv) > 1 else "ab"
wl = sy.workload(640, 480, 27, 27)
rs, ws = torch.from_numpy(wl["render_stack"]).cuda(), torch.from_numpy(wl["warp_stack"]).cuda()
ctx = nmi.NmiContext(640, 480)
if mode in ("0", "1"):  # for PMC runs: one setting only
    ctx.set_option(ctx.OPT_XCD_TILING, int(mode))
    for _ in range(30):
        r = ctx.search_grid(rs, ws)
    assert r[0] == wl["planted"]
    sys.exit(0)
ctx.set_profiling(True)
t = {0: [], 1: []}
for rnd in range(13):
    for v in (0, 1):
        ctx.set_option(ctx.OPT_XCD_TILING, v)
        for _ in range(3):
            r = ctx.search_grid(rs, ws)
            assert r[0] == wl["planted"]
            if rnd: t[v].append(ctx.last_kernel_ms() * 1e3)
for v in (0, 1):
    print(f"xcd_tiling={v}: median {np.median(t[v]):.1f} us  min {np.min(t[v]):.1f} us")
