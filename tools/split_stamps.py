#!/usr/bin/env python3
"""Where a part of the split kernel spends its time: wall_clock64 stamps (100 MHz) at the phase boundaries of every
workgroup of one launch (NMI_OPT_STAMPS).  python tools/split_stamps.py [S Wn [K]]"""
import os, sys
EVENTS = "--events" in sys.argv  # record HIP events around the stamped launch (does the marker ahead of it change the start skew?)
sys.argv = [x for x in sys.argv if x != "--events"]
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import orbslam2_nmi_amd as nmi
from orbslam2_nmi_amd import synthetic as sy

S = int(sys.argv[1]) if len(sys.argv) > 1 else 1
Wn = int(sys.argv[2]) if len(sys.argv) > 2 else 1
K = int(sys.argv[3]) if len(sys.argv) > 3 else 8
MASK = int(sys.argv[4]) if len(sys.argv) > 4 else 3
P = int(sys.argv[5]) if len(sys.argv) > 5 else 1
w, h = 640, 480
wl = sy.workload(w, h, 27, 27, seed=1234)
rs, ws = torch.from_numpy(wl["render_stack"]).cuda()[:S].contiguous(), torch.from_numpy(wl["warp_stack"]).cuda()[:Wn].contiguous()
names = ["start", "zeroed", "hist", "decode", "ticket", "final", "end"]
with nmi.NmiContext(w, h) as ctx:
    ctx.set_option(ctx.OPT_SPLIT, K)
    ctx.set_option(ctx.OPT_SPLIT_PIXELS, P)
    ctx.set_option(ctx.OPT_PHASE_MASK, MASK)
    n_wg = (S * Wn if (S * Wn < 8 and K == 8) else ((S * Wn + 7) // 8) * 8) * K * P
    st = torch.zeros((n_wg, 8), dtype=torch.int64, device="cuda")
    torch.cuda.synchronize()
    for rep in range(5):
        ctx.search_grid(rs, ws)
    ctx.set_option(ctx.OPT_STAMPS, st.data_ptr())
    ctx.set_profiling(EVENTS)
    for rep in range(2):
        st.zero_()
        torch.cuda.synchronize()
        ctx.search_grid(rs, ws)
        ctx.synchronize()
        a = st.cpu().numpy().astype(np.float64)
        live = a[:, 1] > 0
        t0 = a[live, 0].min()
        print(f"launch {rep}: {live.sum()} working workgroups of {n_wg}")
        dur = (a[live, 6] - a[live, 0]) / 100.0
        print(f"  shader clock over the workgroups' lifetimes: {np.mean(a[live, 7] / dur) / 1e3:.2f} GHz (clock64 ticks / wall us)")
        print(f'    start: mean {np.mean(a[live, 0] - t0) / 100:.2f} us  max {np.max(a[live, 0] - t0) / 100:.2f}; by XCD (block % 8): ' + '  '.join(f'{np.mean(a[live, 0][np.nonzero(live)[0] % 8 == x] - t0) / 100:.2f}' for x in range(8)))
        for k in range(1, 7):
            col = a[live, k]
            col = col[col > 0]
            if col.size:
                print(f"  {names[k]:>7}: mean {np.mean(col - t0) / 100:.2f} us  min {np.min(col - t0) / 100:.2f}  max {np.max(col - t0) / 100:.2f}  (n={col.size})")
    ctx.set_option(ctx.OPT_STAMPS, 0)
