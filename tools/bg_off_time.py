#!/usr/bin/env python3
"""Kernel time of one 729-candidate search at 640x480 with the background rule (nmi_prop_BG, NMI.cu:85) on and off, on the
optimistic path (NMI_OPT_HIST_VARIANT 3, default) and on the exact path (1)."""
import sys, numpy as np, torch
sys.path.insert(0, '.')
import orbslam2_nmi_amd as nmi
from orbslam2_nmi_amd import synthetic as sy
wl = sy.workload(640, 480, 27, 27)
rs, ws = torch.from_numpy(wl["render_stack"]).cuda(), torch.from_numpy(wl["warp_stack"]).cuda()
for bg in (True, False):
    for hv in (3, 1):
        with nmi.NmiContext(640, 480, use_bg=bg) as ctx:
            ctx.set_profiling(True); ctx.set_option(ctx.OPT_HIST_VARIANT, hv)
            t = []
            for i in range(20):
                ctx.search_grid(rs, ws); t.append(ctx.last_kernel_ms() * 1e3)
            print(f"use_bg={bg} hist_variant={hv}: {np.median(t[3:]):.1f} us per 729-candidate search", flush=True)
