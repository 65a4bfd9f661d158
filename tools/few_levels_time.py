#!/usr/bin/env python3
"""Per-kernel times of the few-levels path (probe, rank images, scoring, gated fall-back launch) on posterised content.

Run under rocprofv3 and read the kernel trace:
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_few -- python3 tools/few_levels_time.py
"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import orbslam2_nmi_amd as nmi
from orbslam2_nmi_amd import synthetic as sy

W, H, S, Wn = 640, 480, 27, 27
wl = sy.workload(W, H, S, Wn)


def posterise(a, levels):
    q = 256 // levels
    return (a // q * q + q // 2).astype(np.uint8)


ctx = nmi.NmiContext(W, H)
ctx.set_profiling(True)
ctx.set_option(ctx.OPT_FEWLEVELS_BINS, 4096)
for levels in (64, 32, 16, 4):
    rs = torch.from_numpy(posterise(wl["render_stack"], levels)).cuda()
    ws = torch.from_numpy(posterise(wl["warp_stack"], levels)).cuda()
    ratings = torch.empty((Wn, S), dtype=torch.float32, device="cuda")
    for path in (0, 1):
        ctx.set_option(ctx.OPT_CONTENT_PATH, path)
        t = []
        for i in range(12):
            ctx.search_grid(rs, ws, ratings=ratings)
            t.append(ctx.last_kernel_ms() * 1e3)
        print(f"{levels:3d} levels  path {path}: {np.median(t[2:]):7.1f} us per search   {ctx.last_content()}", flush=True)

# fewer than 256 bins on the default (textured) scene: at most `bins` levels per stack
rs, ws = torch.from_numpy(wl["render_stack"]).cuda(), torch.from_numpy(wl["warp_stack"]).cuda()
for bins in (128, 64, 32, 16):
    with nmi.NmiContext(W, H, bins=bins) as c2:
        c2.set_profiling(True)
        for path in (0, 1):
            c2.set_option(c2.OPT_CONTENT_PATH, path)
            t = []
            for i in range(12):
                c2.search_grid(rs, ws)
                t.append(c2.last_kernel_ms() * 1e3)
            print(f"default scene, {bins:3d} bins  path {path}: {np.median(t[2:]):7.1f} us per search   {c2.last_content()}", flush=True)
