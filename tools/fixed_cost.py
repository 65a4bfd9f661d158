#!/usr/bin/env python3
"""Where does the kernel's fixed cost go?  Empty-loop (phase mask 0) kernel time vs grid size and workgroup count."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import orbslam2_nmi_amd as nmi
W, H = 640, 480
ctx = nmi.NmiContext(W, H)
ctx.set_profiling(True)
rng = np.random.default_rng(0)
ws = torch.from_numpy(rng.integers(0, 256, (27, H, W), dtype=np.uint8)).cuda()
rs = torch.from_numpy(rng.integers(0, 256, (27, H, W), dtype=np.uint8)).cuda()
def t(S, Wn, pm, wg=0, n=15):
    ctx.set_option(ctx.OPT_PHASE_MASK, pm); ctx.set_option(ctx.OPT_WORKGROUPS, wg)
    v = []
    for i in range(n + 3):
        ctx.search_grid(rs[:S], ws[:Wn])
        if i >= 3: v.append(ctx.last_kernel_ms() * 1e3)
    return np.median(v)
for (S, Wn) in ((1, 1), (16, 16), (27, 27)):
    for wg in (0, 64, 128):
        print(f"S={S:2d} Wn={Wn:2d} wg={wg:3d}: empty {t(S, Wn, 0, wg):6.1f} us   decode-only {t(S, Wn, 2, wg):6.1f} us   full {t(S, Wn, 3, wg):6.1f} us")
