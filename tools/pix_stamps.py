#!/usr/bin/env python3
"""Where the workgroups of nmi_pix_kernel (mid-size grids: P pixel ranges per candidate) spend their time: wall_clock64
stamps (100 MHz) of every workgroup of one launch (NMI_OPT_STAMPS).  python tools/pix_stamps.py [S Wn [P]]"""
import os, sys
EVENTS = "--events" in sys.argv  # record HIP events around the stamped launch (does the marker ahead of it change the start skew?)
sys.argv = [x for x in sys.argv if x != "--events"]
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import orbslam2_nmi_amd as nmi
from orbslam2_nmi_amd import synthetic as sy

NOISE = "--noise" in sys.argv
sys.argv = [x for x in sys.argv if x != "--noise"]
S = int(sys.argv[1]) if len(sys.argv) > 1 else 9
Wn = int(sys.argv[2]) if len(sys.argv) > 2 else 9
P = int(sys.argv[3]) if len(sys.argv) > 3 else 3
w, h = 640, 480
wl = sy.workload(w, h, 27, 27, seed=1234)
rs, ws = torch.from_numpy(wl["render_stack"]).cuda()[:S].contiguous(), torch.from_numpy(wl["warp_stack"]).cuda()[:Wn].contiguous()
if NOISE:  # uniform noise: no flat regions, every bin equally likely
    g = torch.Generator(device="cuda").manual_seed(1)
    rs = torch.randint(0, 256, rs.shape, dtype=torch.uint8, device="cuda", generator=g)
    ws = torch.randint(0, 256, ws.shape, dtype=torch.uint8, device="cuda", generator=g)
total = S * Wn
with nmi.NmiContext(w, h) as ctx:
    ctx.set_option(ctx.OPT_SPLIT, 1)
    ctx.set_option(ctx.OPT_SPLIT_PIXELS, P)
    st = torch.zeros((total * P, 8), dtype=torch.int64, device="cuda")
    for rep in range(20):
        ctx.search_grid(rs, ws)
    assert ctx.pix_status()["last_launch_ranges"] == P
    ctx.set_profiling(True)
    d = []
    for rep in range(30):
        ctx.search_grid(rs, ws)
        d.append(ctx.last_kernel_ms())
    ctx.set_profiling(False)
    print(f"{S}x{Wn}, {P} ranges: nmi_pix_kernel {np.median(d) * 1e3:.1f} us (HIP events, unstamped)")
    ctx.set_option(ctx.OPT_STAMPS, st.data_ptr())
    ctx.set_profiling(EVENTS)
    for rep in range(3):
        st.zero_()
        torch.cuda.synchronize()
        ctx.search_grid(rs, ws)
        ctx.synchronize()
        a = st.cpu().numpy().astype(np.float64)
        t0 = a[:, 0].min()
        hl, ow = a[: total * (P - 1)], a[total * (P - 1):]
        print(f"launch {rep}: times from the first workgroup's start (us)")
        for name, rows, cols in (("helpers", hl, ["start", "cleared", "hist (barrier)", "masks stored"]),
                                 ("owners", ow, ["start", "cleared", "hist(wave0)", "B1, units in", "", "decoded", "scored", "end"])):
            for k, c in enumerate(cols):
                col = rows[:, k]
                col = col[col > 0]
                if col.size and c:
                    print(f"  {name:>8} {c:>14}: mean {np.mean(col - t0) / 100:6.2f}  min {np.min(col - t0) / 100:6.2f}  max {np.max(col - t0) / 100:6.2f}")
        if rep == 2:
            for qq in range(P - 1):
                rows = hl[qq * total:(qq + 1) * total]
                d = (rows[:, 2] - rows[:, 1]) / 100
                print(f"  helper range {qq + 1}: histogram phase (cleared -> barrier) mean {d.mean():.2f}  min {d.min():.2f}  max {d.max():.2f};  slowest candidates: {np.argsort(-d)[:6].tolist()}")
            d = (ow[:, 3] - ow[:, 1]) / 100
            print(f"  owners: cleared -> B1 mean {d.mean():.2f}  min {d.min():.2f}  max {d.max():.2f};  slowest candidates: {np.argsort(-d)[:6].tolist()}")
            d = (ow[:, 7] - t0) / 100
            print(f"  owners: end, slowest candidates: {np.argsort(-d)[:8].tolist()}  their XCDs {[(int(total * (P - 1) + c)) % 8 for c in np.argsort(-d)[:8]]}")
            dur = (a[:, 2] - a[:, 1]) / 100  # cleared -> hist end (helpers: barrier; owners: wave 0 only)
            blk = np.arange(total * P)
            for qq, name in [(1, "helper 1"), (2, "helper 2"), (0, "owner (wave 0)")][: P if P < 3 else 3]:
                sel = (blk // total == qq - 1) if qq else (blk >= total * (P - 1))
                print(f"  {name:>15} hist by XCD: " + "  ".join(f"{dur[sel & (blk % 8 == x)].mean():.2f}" for x in range(8)))
            start = (a[:, 0] - t0) / 100
            print("  start by XCD (block % 8): " + "  ".join(f"{start[x::8].mean():.2f}" for x in range(8)))
            print("  start by position in the XCD's queue (block // 8), every 4th: " + "  ".join(f"{start[8 * j: 8 * j + 8].mean():.2f}" for j in range(0, (total * P) // 8, 4)))
    ctx.set_option(ctx.OPT_STAMPS, 0)
