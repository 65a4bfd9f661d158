#!/usr/bin/env python3
"""nmi_pix_kernel: kernel time against NMI_OPT_PIX_OWNER_BIAS (pixels the owner adds beyond an equal share) per grid and number
of ranges.  python tools/pix_bias_sweep.py"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import orbslam2_nmi_amd as nmi
from orbslam2_nmi_amd import synthetic as sy

w, h = 640, 480
wl = sy.workload(w, h, 27, 27, seed=1234)
rs, ws = torch.from_numpy(wl["render_stack"]).cuda(), torch.from_numpy(wl["warp_stack"]).cuda()
biases = [0, 16384, 32768, 49152, 65536, 81920, 98304, 131072]
print("grid  ranges | " + " ".join(f"{b // 1024:>5}k" for b in biases))
for S, Wn, P in [(9, 9, 3), (9, 9, 2), (27, 4, 2), (16, 4, 4), (16, 4, 3), (16, 3, 5), (16, 3, 4), (27, 1, 4), (27, 1, 3)]:
    r, v = rs[:S].contiguous(), ws[:Wn].contiguous()
    out = []
    with nmi.NmiContext(w, h) as ctx:
        ctx.set_option(ctx.OPT_SPLIT, 1)
        ctx.set_option(ctx.OPT_SPLIT_PIXELS, P)
        for b in biases:
            ctx.set_option(ctx.OPT_PIX_OWNER_BIAS, b)
            for _ in range(10):
                ctx.search_grid(r, v)
            ctx.set_profiling(True)
            d = []
            for _ in range(40):
                ctx.search_grid(r, v)
                d.append(ctx.last_kernel_ms())
            ctx.set_profiling(False)
            out.append(np.median(d) * 1e3)
    print(f"{S:>2}x{Wn:<2} {P:>6} | " + " ".join(f"{x:6.1f}" for x in out), flush=True)
