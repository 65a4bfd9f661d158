#!/usr/bin/env python3
"""Back-to-back launch rate (throughput mode) with and without the completion protocol (phase-mask bit 4)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import orbslam2_nmi_amd as nmi
from orbslam2_nmi_amd import synthetic as sy
wl = sy.workload(640, 480, 27, 27)
st = torch.cuda.Stream(); torch.cuda.set_stream(st)
rs, ws = torch.from_numpy(wl["render_stack"]).cuda(), torch.from_numpy(wl["warp_stack"]).cuda()
ctx = nmi.NmiContext(640, 480); ctx.set_stream(st.cuda_stream)
keys = torch.zeros(256, dtype=torch.int64, device="cuda")
for rnd in range(3):
    for pm in (3, 19):
        ctx.set_option(ctx.OPT_PHASE_MASK, pm)
        for i in range(20): ctx.search_grid_shard(rs, 0, 27, ws, key_out=keys[i:i+1], blocking=False)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for i in range(200): ctx.search_grid_shard(rs, 0, 27, ws, key_out=keys[i:i+1], blocking=False)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 200 * 1e6
        print(f"round {rnd} phase_mask {pm}: {dt:.1f} us/step  {729/dt:.2f} M evals/s")
