#!/usr/bin/env python3
"""The cost of frame widths that are not multiples of 16 (the histogram phase's 16-byte loads need width % 16 == 0; other widths
take the byte path, nmi_kernels.hip): 729 candidates at 640x480 against 641x480 / 648x480, and KITTI's 1241x376 against
1248x376.  python tools/odd_width_time.py"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import orbslam2_nmi_amd as nmi
from orbslam2_nmi_amd import synthetic as sy

print(f"{'frame':>10} {'form':>8} {'kernel us':>10} {'us / Mpixel-pair':>17}")
for w, h in [(640, 480), (641, 480), (648, 480), (1248, 376), (1241, 376)]:
    wl = sy.workload(w, h, 27, 27, seed=1234)
    rs, ws = torch.from_numpy(wl["render_stack"]).cuda(), torch.from_numpy(wl["warp_stack"]).cuda()
    with nmi.NmiContext(w, h) as ctx:
        for _ in range(5):
            got = ctx.search_grid(rs, ws)
        assert got[0] == wl["planted"], (w, h, got)
        ctx.set_profiling(True)
        d = []
        for _ in range(30):
            ctx.search_grid(rs, ws)
            d.append(ctx.last_kernel_ms())
        ctx.set_profiling(False)
    us = np.median(d) * 1e3
    print(f"{w:>5}x{h:<4} {'aligned' if w % 16 == 0 else 'rows':>8} {us:>10.1f} {us / (729 * w * h / 1e6):>17.3f}", flush=True)
