"""Long randomised parity campaign (tests/fuzz_parity.py) over the kernel selections: default, few-levels first, low few-levels
limit, general kernel only; small and larger grids.  On a GPU box:  python tools/long_fuzz.py [seed offset]"""
import sys, time
OFFSET = int(sys.argv[1]) if len(sys.argv) > 1 else 0
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
                                   ("no split, general kernel only", 400, 104, {C.OPT_SPLIT: 0, C.OPT_CONTENT_PATH: 0}, 7, 8),
                                   ("pixel ranges only, 3 per candidate, grids up to 16 x 16", 400, 107, {C.OPT_SPLIT: 1, C.OPT_SPLIT_PIXELS: 3, C.OPT_CONTENT_PATH: 0}, 7, 16),
                                   ("pixel ranges only, 2 per candidate, owner bias 150 k", 300, 108, {C.OPT_SPLIT: 1, C.OPT_SPLIT_PIXELS: 2, C.OPT_PIX_OWNER_BIAS: 150000, C.OPT_CONTENT_PATH: 0}, 7, 11),
                                   ("pixel ranges only, 5 per candidate, equal shares", 300, 109, {C.OPT_SPLIT: 1, C.OPT_SPLIT_PIXELS: 5, C.OPT_PIX_OWNER_BIAS: 0, C.OPT_CONTENT_PATH: 0}, 7, 7)):
    w = fuzz_parity.run(n, seed=seed + OFFSET, verbose=False, options=opts, kinds=kinds, max_side=side)
    print(f"{name} (seed {seed + OFFSET}): {n} cases, worst |score error| {w:.1e}, {time.time() - t0:.0f} s", flush=True)
