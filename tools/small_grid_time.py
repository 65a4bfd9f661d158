#!/usr/bin/env python3
"""Kernel time and blocking-call time of small and mid-size grids (1 ... 243 candidates at 640x480) per kernel selection:
0 = one workgroup per candidate (nmi_grid_kernel), -1 = automatic, K = row parts (nmi_split_kernel), (1, P) = P pixel
ranges per candidate (nmi_pix_kernel); plus the per-call rate of nmi_eval_pair (the reference's unchanged call site).
Run on a GPU box: python tools/small_grid_time.py [--quick]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import orbslam2_nmi_amd as nmi
from orbslam2_nmi_amd import synthetic as sy

quick = "--quick" in sys.argv
w, h = 640, 480
wl = sy.workload(w, h, 27, 27, seed=1234)
rs, ws = torch.from_numpy(wl["render_stack"]).cuda(), torch.from_numpy(wl["warp_stack"]).cuda()
stream = torch.cuda.Stream()
torch.cuda.set_stream(stream)
grids = [(1, 1), (3, 1), (9, 1), (27, 1), (9, 4), (16, 3), (16, 4), (9, 9), (27, 4), (27, 7), (27, 9)]
if quick:
    grids = [(27, 1), (16, 4), (9, 9), (27, 4), (27, 9)]
print(f"{'grid':>8} {'mode':>8} {'kernel us':>10} {'call us':>9} {'evals/s':>10}")
for S, Wn in grids:
    r, v = rs[:S].contiguous(), ws[:Wn].contiguous()
    ref = None
    total = S * Wn
    modes = [0, -1] + ([8, 4] if total <= 64 and not quick else []) + [(1, p) for p in (2, 3, 4, 5) if total * p <= 256]
    for mode in modes:
        with nmi.NmiContext(w, h) as ctx:
            ctx.set_stream(stream.cuda_stream)
            if isinstance(mode, tuple):
                ctx.set_option(ctx.OPT_SPLIT, mode[0])
                ctx.set_option(ctx.OPT_SPLIT_PIXELS, mode[1])
            else:
                ctx.set_option(ctx.OPT_SPLIT, mode)
                if mode > 0:
                    ctx.set_option(ctx.OPT_SPLIT_PIXELS, 1)
            for _ in range(20):
                got = ctx.search_grid(r, v)
            ref = ref or got
            assert got == ref, (S, Wn, mode, got, ref)
            ctx.set_profiling(True)
            d = []
            for _ in range(50):
                ctx.search_grid(r, v)
                d.append(ctx.last_kernel_ms())
            ctx.set_profiling(False)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(300):
                ctx.search_grid(r, v)
            dt = (time.perf_counter() - t0) / 300
            name = f"1x{mode[1]}" if isinstance(mode, tuple) else str(mode)
            used = ctx.pix_status()["last_launch_ranges"]
            print(f"{S:>4}x{Wn:<3} {name:>8} {np.median(d) * 1e3:>10.1f} {dt * 1e6:>9.1f} {S * Wn / dt:>10.0f}" + (f"   ({used} pixel ranges)" if used and mode == -1 else ""), flush=True)
if not quick:
    for mode in (0, -1):
        with nmi.NmiContext(w, h) as ctx:
            ctx.set_stream(stream.cuda_stream)
            ctx.set_option(ctx.OPT_SPLIT, mode)
            for wait in (0, 1):
                ctx.set_option(ctx.OPT_WAIT_MODE, wait)
                for _ in range(50):
                    ctx.eval_pair(rs[0], ws[0])
                t0 = time.perf_counter()
                for i in range(2000):
                    ctx.eval_pair(rs[i % 27], ws[(i * 7) % 27])
                dt = (time.perf_counter() - t0) / 2000
                print(f"eval_pair split={mode} wait={wait}: {dt * 1e6:.1f} us per call = {1 / dt:.0f} evals/s (python ctypes loop)", flush=True)
