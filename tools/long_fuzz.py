"""Long randomised parity campaign (tests/fuzz_parity.py) over the kernel selections: default, few-levels first, low few-levels
limit, general kernel only; small and larger grids.  On a GPU box:  python tools/long_fuzz.py"""
import sys, time
sys.path.insert(0, 'tests'); sys.path.insert(0, '.')
import fuzz_parity
import orbslam2_nmi_amd as nmi
C = nmi.NmiContext
t0 = time.time()
for name, n, seed, opts, kinds, side in (("default options", 1200, 101, None, 7, 8),
                                   ("few-levels first, no split", 1200, 102, {C.OPT_SPLIT: 0, C.OPT_CONTENT_PATH: 1}, 7, 8),
                                   ("default options, grids up to 16 x 16", 500, 105, None, 7, 16),
                                   ("few-levels first, grids up to 16 x 16", 500, 106, {C.OPT_CONTENT_PATH: 1}, 7, 16),
                                   ("few-levels first, limit 300 bins", 600, 103, {C.OPT_SPLIT: 0, C.OPT_CONTENT_PATH: 1, C.OPT_FEWLEVELS_BINS: 300}, 7, 8),
                                   ("no split, general kernel only", 400, 104, {C.OPT_SPLIT: 0, C.OPT_CONTENT_PATH: 0}, 7, 8)):
    w = fuzz_parity.run(n, seed=seed, verbose=False, options=opts, kinds=kinds, max_side=side)
    print(f"{name}: {n} cases, worst |score error| {w:.1e}, {time.time() - t0:.0f} s", flush=True)
