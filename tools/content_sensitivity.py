#!/usr/bin/env python3
"""Kernel time of one 729-candidate search at 640x480 as a function of image CONTENT.

The scoring kernel is bound by LDS atomics, whose cost depends on how many lanes of a wave hit the same bank or
the same bin -- i.e. on the images.  This tool measures the default synthetic workload next to content chosen to
stress that: uniform noise (no locality), posterised scenes (few bins), large flat regions, constant images.
Timing only (the rating table is requested, as a caller would): parity on this kind of content is the business of
tests/test_gpu_parity.py (test_flat_regions_*, test_fuzz_campaign_against_the_oracle).
"""
import os, sys, json
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import orbslam2_nmi_amd as nmi
from orbslam2_nmi_amd import synthetic as sy

W, H, S, Wn = 640, 480, 27, 27
wl = sy.workload(W, H, S, Wn)
rng = np.random.default_rng(7)


def posterise(a, levels):
    q = 256 // levels
    return (a // q * q + q // 2).astype(np.uint8)


def flat_top(a, frac, value):
    b = a.copy()
    b[..., : int(H * frac), :] = value
    return b


cases = {
    "default (smooth scene + noise)": (wl["render_stack"], wl["warp_stack"]),
    "uniform noise": (rng.integers(0, 256, (S, H, W), dtype=np.uint8), rng.integers(0, 256, (Wn, H, W), dtype=np.uint8)),
    "posterised to 64 levels": (posterise(wl["render_stack"], 64), posterise(wl["warp_stack"], 64)),
    "posterised to 32 levels": (posterise(wl["render_stack"], 32), posterise(wl["warp_stack"], 32)),
    "posterised to 16 levels": (posterise(wl["render_stack"], 16), posterise(wl["warp_stack"], 16)),
    "posterised to 4 levels": (posterise(wl["render_stack"], 4), posterise(wl["warp_stack"], 4)),
    # rows are bottom-up in the render and top-down in the warp: flatten the same scene rows in both
    "30% flat (render 255 over frame 255)": (flat_top(wl["render_stack"][:, ::-1], 0.3, 255)[:, ::-1].copy(),
                                              flat_top(wl["warp_stack"], 0.3, 255)),
    "60% flat (render 255 over frame 0)": (flat_top(wl["render_stack"][:, ::-1], 0.6, 255)[:, ::-1].copy(),
                                            flat_top(wl["warp_stack"], 0.6, 0)),
    "constant images": (np.full((S, H, W), 255, np.uint8), np.full((Wn, H, W), 17, np.uint8)),
}

ctx = nmi.NmiContext(W, H)
ctx.set_profiling(True)
out = []
dev = {}
for name, (rs_h, ws_h) in cases.items():
    dev[name] = (torch.from_numpy(np.ascontiguousarray(rs_h)).cuda(), torch.from_numpy(np.ascontiguousarray(ws_h)).cuda())
ratings = torch.empty((Wn, S), dtype=torch.float32, device="cuda")


def searches(name, n):
    rs, ws = dev[name]
    t = []
    for _ in range(n):
        ctx.search_grid(rs, ws, ratings=ratings)
        t.append(ctx.last_kernel_ms() * 1e3)
    return t


for name in cases:
    t = searches(name, 60)  # default options: the path follows the content (NMI_OPT_CONTENT_PATH -1)
    us, first = float(np.median(t[-10:])), float(t[0])
    info = ctx.last_content()
    path = "few-levels (%d x %d)" % (info["nr"], info["nw"]) if info["few_levels"] else "general"
    out.append({"content": name, "kernel_us": round(us, 1), "evals_per_s": round(S * Wn / us * 1e6), "path": path,
                "first_search_us": round(first, 1)})
    print(f"{name:42s} {us:8.1f} us  {S * Wn / us:6.2f} M evals/s   {path}   (first search after the change of content {first:.1f} us)", flush=True)

# Transitions: a stream whose content changes.  How many searches, and how many microseconds, until the steady path's time
# (within 10 % of the content's steady-state time above) is reached?  Every general search is a content probe (its last
# workgroup posts the bins it found), every few-levels search carries its own probe: one search either way.
steady = {o["content"]: o["kernel_us"] for o in out}
transitions = []
print("\ntransitions (searches until within 10 % of the steady time; times of the first searches after the switch, us)")
for a, b in (("default (smooth scene + noise)", "posterised to 4 levels"), ("posterised to 4 levels", "default (smooth scene + noise)"),
             ("default (smooth scene + noise)", "posterised to 16 levels"), ("posterised to 16 levels", "default (smooth scene + noise)"),
             ("uniform noise", "constant images"), ("constant images", "uniform noise")):
    searches(a, 20)
    t = searches(b, 12)
    slow = next((i for i, v in enumerate(t) if v <= 1.1 * steady[b]), len(t))
    extra = float(sum(v - steady[b] for v in t[:slow]))
    transitions.append({"from": a, "to": b, "slow_searches": slow, "extra_us": round(extra, 1), "first_us": [round(v, 1) for v in t[:4]]})
    print(f"{a:34s} -> {b:34s} {slow} slow search(es), {extra:7.1f} us lost   first four: {[round(v, 1) for v in t[:4]]}", flush=True)
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump({"steady": out, "transitions": transitions}, open(os.path.join(ROOT, "gpurun_out", "content_sensitivity.json"), "w"), indent=1)
