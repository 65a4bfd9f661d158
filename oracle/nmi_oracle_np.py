"""numpy twin of oracle/nmi_oracle.c -- TEST INFRASTRUCTURE ONLY ("parity unpinned", see the C header).

Written independently of the C file (vectorised, different loop structure) so that the two
restatements check each other.  Used by tests/ and by tests/golden/make_golden.py; never by the
product.  Reference citations are relative to the reference repository root.
"""
import numpy as np

BINS = 256
MODE_ENMI = 0  # Thirdparty/CUDA_Functions/kernel.cuh:22
MODE_SUC = 1   # Thirdparty/CUDA_Functions/kernel.cuh:23


def joint_hist(render, warped, shift=0, use_bg=True, render_bottom_up=True):
    """Thirdparty/CUDA_Functions/NMI.cu:42-49,79-87 -> (joint[256,256], hist1[256], hist2[256]) as uint32.

    render/warped: uint8 [H, W].  Row y of the warped frame meets row H-1-y of the render when
    render_bottom_up (NMI.cu:82); a pixel counts iff use_bg or both raw intensities are non-zero (:85).
    """
    render = np.asarray(render, dtype=np.uint8)
    warped = np.asarray(warped, dtype=np.uint8)
    assert render.shape == warped.shape and render.ndim == 2
    r = render[::-1] if render_bottom_up else render
    d1 = r.reshape(-1).astype(np.int64)
    d2 = warped.reshape(-1).astype(np.int64)
    if not use_bg:
        keep = (d1 != 0) & (d2 != 0)
        d1, d2 = d1[keep], d2[keep]
    d1 >>= shift
    d2 >>= shift
    joint = np.bincount(d1 * BINS + d2, minlength=BINS * BINS).astype(np.uint32).reshape(BINS, BINS)
    return joint, joint.sum(axis=1, dtype=np.uint32), joint.sum(axis=0, dtype=np.uint32)


def bin_terms(counts, length):
    """ComputeEntropyKernel, NMI.cu:242-263, in fp32: 0 for empty bins, else p*log2f(p), p = c/len."""
    c = np.asarray(counts)
    p = c.astype(np.float32) / np.float32(length)
    with np.errstate(divide="ignore", invalid="ignore"):
        t = p * np.log2(p, dtype=np.float32)
    return np.where(c == 0, np.float32(0), t).astype(np.float32)


def tree256(a):
    """Stride-halving fp32 tree over the last axis (length 256): NMI.cu:276-284 / :299-307."""
    a = np.array(a, dtype=np.float32)
    n = BINS // 2
    while n >= 1:
        a[..., :n] = a[..., :n] + a[..., n:2 * n]
        n //= 2
    return a[..., 0]


def score(a1, a2, a3, mode=MODE_SUC):
    """AddVectorPairwiseKernel, NMI.cu:342-362 (intended value; the reference has no grid sync there)."""
    a1, a2, a3 = np.float32(a1), np.float32(a2), np.float32(a3)
    if a1 == 0 and a2 == 0 and a3 == 0:
        return np.float32(0)
    with np.errstate(divide="ignore", invalid="ignore"):
        if mode == MODE_ENMI:
            return np.float32(((-a1) + (-a2)) / (-a3))
        if mode == MODE_SUC:
            return np.float32(np.float32(2) * (np.float32(1) - ((-a3) / ((-a1) + (-a2)))))
    return np.float32(-1)


def score_from_hist(joint, hist1, hist2, length, mode=MODE_SUC, return_sums=False):
    """kernel.cu:83-95: per-bin terms, joint rows first (row = fixed render intensity), then three trees."""
    a1 = tree256(bin_terms(hist1, length))
    a2 = tree256(bin_terms(hist2, length))
    a3 = tree256(tree256(bin_terms(np.asarray(joint).reshape(BINS, BINS), length)))
    s = score(a1, a2, a3, mode)
    return (s, (a1, a2, a3)) if return_sums else s


def eval_pair(render, warped, shift=0, use_bg=True, render_bottom_up=True, mode=MODE_SUC):
    """CUDAF::NMIWithCuda_noMask, kernel.cu:49-114 (arithmetic only)."""
    j, h1, h2 = joint_hist(render, warped, shift, use_bg, render_bottom_up)
    return score_from_hist(j, h1, h2, render.shape[0] * render.shape[1], mode)


def find_max(ratings):
    """helperFunctions.cpp:50-103 + Tracking.cc:1952: strict '>' from 0, first cell equal to the max; -1 if none."""
    r = np.asarray(ratings, dtype=np.float32).reshape(-1)
    m = np.float32(0)
    for v in r:
        if v > m:
            m = v
    hit = np.nonzero(r == m)[0]
    return (int(hit[0]), np.float32(r[hit[0]])) if hit.size else (-1, np.float32(0))


def search_grid(render_stack, warp_stack, **kw):
    """Tracking.cc:1879-1902 + find_max: ratings[w, s] and the winner's linear index w*S+s."""
    S, Wn = len(render_stack), len(warp_stack)
    ratings = np.zeros((Wn, S), dtype=np.float32)
    for w in range(Wn):
        for s in range(S):
            ratings[w, s] = eval_pair(render_stack[s], warp_stack[w], **kw)
    idx, best = find_max(ratings)
    return ratings, idx, best
