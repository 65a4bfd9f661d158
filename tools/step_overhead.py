#!/usr/bin/env python3
"""Per-step host overhead of nmi_search_grid: wall time per blocking call vs HIP-event kernel time, for the two
result paths (NMI_OPT_RESULT_PATH 1 = mailbox poll, 0 = hipMemcpyAsync + hipStreamSynchronize)."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import orbslam2_nmi_amd as nmi
from orbslam2_nmi_amd import synthetic as sy

wl = sy.workload(640, 480, 27, 27)
rs, ws = torch.from_numpy(wl["render_stack"]).cuda(), torch.from_numpy(wl["warp_stack"]).cuda()
ctx = nmi.NmiContext(640, 480)
for rnd in range(3):
    for path in (1, 0):
        ctx.set_option(ctx.OPT_RESULT_PATH, path)
        for _ in range(20):
            r = ctx.search_grid(rs, ws)
        assert r[0] == wl["planted"], r
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(200):
            ctx.search_grid(rs, ws)
        torch.cuda.synchronize()
        wall = (time.perf_counter() - t0) / 200 * 1e6
        ctx.set_profiling(True)
        k = []
        for _ in range(50):
            ctx.search_grid(rs, ws)
            k.append(ctx.last_kernel_ms() * 1e3)
        ctx.set_profiling(False)
        print(f"round {rnd} result_path {path}: wall {wall:7.1f} us/step, kernel {np.mean(k):7.1f} us, overhead {wall - np.mean(k):6.1f} us, "
              f"{729 / wall:6.2f} M evals/s")
