#!/usr/bin/env python3
"""Kernel ablation on one GPU, all variants interleaved in one process (HIP-event time of the grid kernel).

  python tools/ablate.py [--rounds 10] [--data smooth|uniform|constant|all]

Prints median / min kernel time per (data, variant) for the config-2 grid (27 x 27 candidates at 640x480).
Variants: histogram wrap handling (NMI_OPT_HIST_VARIANT) x phases executed (NMI_OPT_PHASE_MASK).
Variants 0 / 2 / 4 exist only in the ablation build of the library:
  python -m orbslam2_nmi_amd.build --ablations && NMI_HIP_LIBRARY=orbslam2_nmi_amd/lib/libnmi_hip_ablate.so python tools/ablate.py
(with the shipped library they are skipped).
"""
import argparse
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rounds", type=int, default=10)
    ap.add_argument("--data", default="all")
    ap.add_argument("--width", type=int, default=640)
    ap.add_argument("--height", type=int, default=480)
    ap.add_argument("--S", type=int, default=27)
    ap.add_argument("--Wn", type=int, default=27)
    ap.add_argument("--json", default=None)
    args = ap.parse_args()
    import torch

    import orbslam2_nmi_amd as nmi
    from orbslam2_nmi_amd import synthetic as sy

    W, H, S, Wn = args.width, args.height, args.S, args.Wn
    datasets = {}
    if args.data in ("smooth", "all"):
        wl = sy.workload(W, H, S, Wn)
        datasets["smooth"] = (wl["render_stack"], wl["warp_stack"])
    if args.data in ("uniform", "all"):
        rng = np.random.default_rng(1236)
        datasets["uniform"] = (rng.integers(0, 256, (S, H, W), dtype=np.uint8), rng.integers(0, 256, (Wn, H, W), dtype=np.uint8))
    if args.data in ("constant", "all"):
        datasets["constant"] = (np.full((S, H, W), 255, np.uint8), np.zeros((Wn, H, W), np.uint8))
    dev = {k: (torch.from_numpy(r).cuda(), torch.from_numpy(w).cuda()) for k, (r, w) in datasets.items()}

    variants = [("hist1 full", 1, 3), ("hist2 full", 2, 3), ("hist3 full", 3, 3), ("hist4 pipelined", 4, 3), ("ws hist+drain", 4, 1), ("ws math+drain", 4, 2), ("ws hist only", 4, 5),
                ("ws math only", 4, 6), ("ws drain only", 4, 0), ("ws empty", 4, 4),
                ("hist1 hist-only", 1, 1), ("hist2 hist-only", 2, 1),
                ("decode-only", 3, 2), ("empty loop", 3, 0)]
    ctx = nmi.NmiContext(W, H)
    ctx.set_profiling(True)
    usable = []
    for v in variants:  # the shipped library has variants 1 and 3 only
        try:
            ctx.set_option(ctx.OPT_HIST_VARIANT, v[1])
            usable.append(v)
        except nmi.NmiError:
            print(f"skipping '{v[0]}': variant {v[1]} is not in this build of the library (see the header of this file)")
    variants = usable
    times = {(d, v[0]): [] for d in dev for v in variants}
    for rnd in range(args.rounds + 1):
        for d, (rs, ws) in dev.items():
            for name, hv, pm in variants:
                ctx.set_option(ctx.OPT_HIST_VARIANT, hv)
                ctx.set_option(ctx.OPT_PHASE_MASK, pm)
                ctx.search_grid(rs, ws)
                if rnd:  # round 0 is warm-up
                    times[(d, name)].append(ctx.last_kernel_ms() * 1e3)
    out = {}
    print(f"{'data':10s} {'variant':18s} {'median us':>10s} {'min us':>10s}  (grid {S}x{Wn} at {W}x{H})")
    for (d, name), t in times.items():
        print(f"{d:10s} {name:18s} {np.median(t):10.1f} {np.min(t):10.1f}")
        out[f"{d}/{name}"] = {"median_us": float(np.median(t)), "min_us": float(np.min(t))}
    if args.json:
        with open(args.json, "w") as f:
            json.dump(out, f, indent=1)


if __name__ == "__main__":
    main()
