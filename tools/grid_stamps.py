#!/usr/bin/env python3
"""Where nmi_grid_kernel's time goes on a grid of S x Wn candidates (one workgroup per candidate): wall_clock64 stamps
(100 MHz) at the phase boundaries of every workgroup's FIRST candidate (NMI_OPT_STAMPS -> nmi_grid_kernel_stamped).
python tools/grid_stamps.py [S Wn]"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import orbslam2_nmi_amd as nmi
from orbslam2_nmi_amd import synthetic as sy

NOISE = "--noise" in sys.argv   # uniform noise instead of the benchmark's frames: no flat regions, no two lanes on one bin
sys.argv = [x for x in sys.argv if x != "--noise"]
S = int(sys.argv[1]) if len(sys.argv) > 1 else 9
Wn = int(sys.argv[2]) if len(sys.argv) > 2 else 9
w, h = 640, 480
wl = sy.workload(w, h, 27, 27, seed=1234)
rs, ws = torch.from_numpy(wl["render_stack"]).cuda()[:S].contiguous(), torch.from_numpy(wl["warp_stack"]).cuda()[:Wn].contiguous()
if NOISE:
    g = torch.Generator(device="cuda").manual_seed(1)
    rs = torch.randint(0, 256, rs.shape, dtype=torch.uint8, device="cuda", generator=g)
    ws = torch.randint(0, 256, ws.shape, dtype=torch.uint8, device="cuda", generator=g)
names = ["start", "cleared", "hist(wave0)", "B1", "decoded(B2)", "scored", "end"]
with nmi.NmiContext(w, h) as ctx:
    ctx.set_option(ctx.OPT_SPLIT, 0)
    n_wg = min(S * Wn, 256)
    st = torch.zeros((n_wg * 24,), dtype=torch.int64, device="cuda")   # [n_wg][8] phase stamps, then [n_wg][16]: every wavefront's last pixel
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for rep in range(20):
        ctx.search_grid(rs, ws)
    ctx.set_profiling(True)
    d = []
    for rep in range(20):
        ctx.search_grid(rs, ws)
        d.append(ctx.last_kernel_ms())
    ctx.set_profiling(False)
    print(f"{S}x{Wn}: nmi_grid_kernel {np.median(d) * 1e3:.1f} us (HIP events, unstamped)")
    ctx.set_option(ctx.OPT_STAMPS, st.data_ptr())
    for rep in range(3):
        st.zero_()
        torch.cuda.synchronize()
        ctx.set_profiling(True)
        ctx.search_grid(rs, ws)
        ctx.synchronize()
        ms = ctx.last_kernel_ms()
        ctx.set_profiling(False)
        raw = st.cpu().numpy().astype(np.float64)
        a = raw[: n_wg * 8].reshape(n_wg, 8)
        wv = raw[n_wg * 8:].reshape(n_wg, 16)
        live = a[:, 1] > 0
        t0 = a[live, 0].min()
        print(f"launch {rep}: {live.sum()} workgroups, stamped kernel {ms * 1e3:.1f} us by HIP events; times from the first workgroup's start")
        for k in range(0, 7):
            col = a[live, k]
            col = col[col > 0]
            if col.size:
                print(f"  {names[k]:>12}: mean {np.mean(col - t0) / 100:6.2f} us  min {np.min(col - t0) / 100:6.2f}  max {np.max(col - t0) / 100:6.2f}  (n={col.size})")
        rel = (wv[live] - a[live, 1:2]) / 100   # every wavefront's last pixel, from its workgroup's "cleared" stamp
        print("  wavefronts' pixel loops end (us after the clear), mean over workgroups: " + " ".join(f"{x:.1f}" for x in rel.mean(axis=0)))
        srt = np.sort(rel, axis=1)
        print("  ... sorted within each workgroup (1st ... 16th to finish):            " + " ".join(f"{x:.1f}" for x in srt.mean(axis=0)))
        per = (a[live, 1:7] - a[live, 0:6]) / 100
        print("  per-workgroup phase lengths (mean): " + "  ".join(f"{names[k + 1]} {per[:, k].mean():.2f}" for k in range(6)))
    ctx.set_option(ctx.OPT_STAMPS, 0)
