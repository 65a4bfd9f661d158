#!/usr/bin/env python3
"""N blocking nmi_eval_pair calls at 640x480 (for rocprofv3 runs of the per-candidate path).  python tools/pair_loop.py [N] [split]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import orbslam2_nmi_amd as nmi
from orbslam2_nmi_amd import synthetic as sy
n = int(sys.argv[1]) if len(sys.argv) > 1 else 200
split = int(sys.argv[2]) if len(sys.argv) > 2 else -1
wl = sy.workload(640, 480, 3, 3, seed=1234)
rs, ws = torch.from_numpy(wl["render_stack"]).cuda(), torch.from_numpy(wl["warp_stack"]).cuda()
import time
mask = int(sys.argv[3]) if len(sys.argv) > 3 else 3
with nmi.NmiContext(640, 480) as ctx:
    ctx.set_option(ctx.OPT_SPLIT, split)
    ctx.set_option(ctx.OPT_PHASE_MASK, mask)
    for rep in range(3):
        t0 = time.perf_counter()
        for i in range(n):
            ctx.eval_pair(rs[i % 3], ws[(i // 3) % 3])
        dt = (time.perf_counter() - t0) / n
        print(f"mask {mask}: {dt * 1e6:.1f} us per call ({1 / dt:.0f} evals/s, python ctypes loop)")
print("done")
