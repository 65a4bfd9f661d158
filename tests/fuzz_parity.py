#!/usr/bin/env python3
"""Randomised parity campaign on the GPU: grids of random shape and content against the C oracle.

Content is drawn to reach every branch of the histogram phase: textured, uniform noise, posterised (bins above 65,535),
flat bands / blocks in one or both stacks (folding, side counters, more flat pairs than side counters), quantised to a
random set of levels (kinds=7; with NMI_OPT_SPLIT 0 and NMI_OPT_CONTENT_PATH 1 these go down the few-levels path), zero regions
with the background rule off, reduced bin counts, both render orientations.  For every case the whole rating table must
EQUAL the oracle's (rounded term mode: same fp32 operations in the same order) and the winner (index and score) must be
the oracle's; the libm-mode oracle (log2f as written) is kept as a <= 1e-5 cross-check.  Test infrastructure (it calls the oracle):
imported by tests/test_gpu_parity.py; for longer campaigns run  python tests/fuzz_parity.py [cases] [seed]  on a GPU box.
"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import orbslam2_nmi_amd as nmi
from orbslam2_nmi_amd import synthetic as sy
from oracle import binding as ob

SHAPES = [(640, 480), (848, 480), (320, 240), (960, 540), (64, 48), (100, 75), (333, 100), (32, 2), (1280, 720)]


def run(cases=200, seed=1, verbose=True, options=None, kinds=6, max_side=8):
    rng = np.random.default_rng(seed)

    def content(kind, n, h, w):
        if kind == 0:
            return rng.integers(0, 256, (n, h, w), dtype=np.uint8)
        if kind == 1:  # smooth + noise
            B = sy.scene(max(w, 32), max(h, 32), int(rng.integers(1 << 30)))[:h, :w]
            return np.clip(B[None].astype(np.int16) + rng.integers(-12, 13, (n, h, w)), 0, 255).astype(np.uint8)
        if kind == 2:  # posterised
            lv = int(rng.choice([2, 4, 16]))
            return (rng.integers(0, lv, (n, h, w)) * (255 // (lv - 1))).astype(np.uint8)
        if kind == 6:  # textured content quantised to 2..70 unevenly spaced levels (the few-levels path's home ground)
            lv = int(rng.integers(2, 71))
            levels = np.sort(rng.choice(256, lv, replace=False)).astype(np.uint8)
            return levels[(content(1, n, h, w).astype(np.int32) * lv) >> 8]
        a = content(int(rng.integers(0, 2)), n, h, w)
        if kind == 3:  # horizontal flat bands
            for i in range(n):
                for _ in range(int(rng.integers(1, 14))):
                    y0 = int(rng.integers(0, h)); y1 = min(h, y0 + int(rng.integers(1, max(2, h // 2))))
                    a[i, y0:y1] = rng.choice([0, 255, int(rng.integers(0, 256))])
            return a
        if kind == 4:  # flat blocks
            for i in range(n):
                for _ in range(int(rng.integers(1, 6))):
                    y0, x0 = int(rng.integers(0, h)), int(rng.integers(0, w))
                    a[i, y0:y0 + int(rng.integers(1, h)), x0:x0 + int(rng.integers(1, w))] = rng.choice([0, 255, 128])
            return a
        return np.full((n, h, w), int(rng.integers(0, 256)), np.uint8)  # constant

    t0 = time.time()
    worst = 0.0
    for c in range(cases):
        w, h = SHAPES[int(rng.integers(len(SHAPES)))]
        S, Wn = int(rng.integers(1, max_side + 1)), int(rng.integers(1, max_side + 1))
        kr, kw = int(rng.integers(0, kinds)), int(rng.integers(0, kinds))
        rs, ws = content(kr, S, h, w), content(kw, Wn, h, w)
        if rng.random() < 0.3 and kr >= 3:  # the same flat rows in both stacks: flat-over-flat pairs
            ws[:, : h // 3] = rs[0, : h // 3][None] if rng.random() < 0.5 else 255
        bins = int(rng.choice([256, 256, 64, 16]))
        bg, bu, mode = bool(rng.random() < 0.8), bool(rng.random() < 0.7), int(rng.integers(0, 2))
        shift = {256: 0, 64: 2, 16: 4}[bins]
        with ob.rounded():
            ref, ibest, vbest = ob.search_grid(rs, ws, shift=shift, use_bg=bg, render_bottom_up=bu, mode=mode, threads=8)
        libm, _, _ = ob.search_grid(rs, ws, shift=shift, use_bg=bg, render_bottom_up=bu, mode=mode, threads=8)
        with nmi.NmiContext(w, h, bins=bins, mode=mode, use_bg=bg, render_bottom_up=bu) as ctx:
            for opt, val in (options or {}).items():
                ctx.set_option(opt, val)
            ratings = torch.empty((Wn, S), dtype=torch.float32, device="cuda")
            idx, val = ctx.search_grid(torch.from_numpy(rs).cuda(), torch.from_numpy(ws).cuda(), ratings=ratings)
            got = ratings.cpu().numpy()
        same = (got.view(np.uint32) == ref.view(np.uint32)) | (np.isnan(got) & np.isnan(ref))
        err = 0.0 if same.all() else float(np.nanmax(np.abs(got - ref)))
        worst = max(worst, err)
        tag = f"case {c} (seed {seed}): {w}x{h} S={S} Wn={Wn} kinds=({kr},{kw}) bins={bins} bg={bg} bu={bu} mode={mode}"
        assert same.all(), (tag, err)
        assert (idx, val) == (ibest, vbest), (tag, idx, ibest, val, vbest)
        fin = np.isfinite(libm)
        assert np.abs(got[fin] - libm[fin]).max(initial=0.0) <= 1e-5 * max(1.0, float(np.abs(libm[fin]).max(initial=0.0))), tag
        if verbose and c % 20 == 19:
            print(f"{c + 1} cases ok, worst |score error| {worst:.2e}, {time.time() - t0:.0f} s", flush=True)
    return worst


if __name__ == "__main__":
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    sd = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    print(f"all {n} cases agree with the oracle; worst |score error| {run(n, sd):.2e}")
