"""Regenerates tests/golden/*.npz.

The reference (gsanya/orbslam2_NMI) holds no golden vectors for the NMI path, and its
CUDA implementation cannot run here, so these fixtures are produced by this repository's own C restatement
(oracle/nmi_oracle.c) and cross-checked against the independent numpy twin before they are written.
They pin the oracle against accidental change and give the GPU tests fixed, committed expectations;
they are NOT outputs of the reference ("parity unpinned").

Run from the repository root:  python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import binding as oc  # noqa: E402
from oracle import nmi_oracle_np as onp  # noqa: E402
from orbslam2_nmi_amd import synthetic as sy  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))
W, H = 64, 48


def pairs():
    rng = np.random.default_rng(20211004)
    B = sy.scene(W, H, 7)
    F = sy.camera_frame(B, 8)
    uni_a = rng.integers(0, 256, (H, W), dtype=np.uint8)
    uni_b = rng.integers(0, 256, (H, W), dtype=np.uint8)
    holes = F.copy()
    holes[rng.random((H, W)) < 0.2] = 0
    rholes = B.copy()
    rholes[:, :9] = 0
    two = (rng.random((H, W)) < 0.5).astype(np.uint8) * 200
    return {
        "smooth": (B, F),
        "uniform": (uni_a, uni_b),
        "identical": (B, B[::-1].copy()),       # identical after the bottom-up flip
        "constant": (np.full((H, W), 255, np.uint8), np.zeros((H, W), np.uint8)),
        "const_vs_smooth": (np.full((H, W), 17, np.uint8), F),
        "zeros_bg": (rholes, holes),
        "two_level": (two, two[::-1].copy() // 2),
    }


def _rounded(fn):
    with oc.rounded():
        return fn()


def main():
    out = {}
    names = []
    for name, (r, w) in pairs().items():
        names.append(name)
        out[f"{name}/render"] = r
        out[f"{name}/warped"] = w
        for bg in (1, 0):
            for bu in (1, 0):
                j, h1, h2 = oc.joint_hist(r, w, 0, bool(bg), bool(bu))
                jn, h1n, h2n = onp.joint_hist(r, w, 0, bool(bg), bool(bu))
                assert (j == jn).all() and (h1 == h1n).all() and (h2 == h2n).all()
                nz = np.flatnonzero(j)
                tag = f"{name}/bg{bg}_bu{bu}"
                out[f"{tag}/joint_idx"] = nz.astype(np.int32)
                out[f"{tag}/joint_cnt"] = j.reshape(-1)[nz].astype(np.uint32)
                out[f"{tag}/hist_render"] = h1
                out[f"{tag}/hist_warped"] = h2
                for mode in (0, 1):
                    s, sums = oc.score_from_hist(j, h1, h2, W * H, mode)
                    sn = onp.score_from_hist(j, h1, h2, W * H, mode)
                    assert abs(float(s) - float(sn)) <= 1e-6 * max(1.0, abs(float(s))), (tag, mode, s, sn)
                    out[f"{tag}/score_mode{mode}"] = np.float32(s)
                    with oc.rounded():  # per-bin log2 correctly rounded: what the GPU tests compare with ==
                        sr, sums_r = oc.score_from_hist(j, h1, h2, W * H, mode)
                    out[f"{tag}/score_mode{mode}_rounded"] = np.float32(sr)
                out[f"{tag}/sums"] = sums
                out[f"{tag}/sums_rounded"] = sums_r
                out[f"{tag}/score64_bins64"] = np.float32(oc.eval_pair(r, w, 2, bool(bg), bool(bu), 1))
                with oc.rounded():
                    out[f"{tag}/score64_bins64_rounded"] = np.float32(oc.eval_pair(r, w, 2, bool(bg), bool(bu), 1))
    out["names"] = np.array(names)
    np.savez_compressed(os.path.join(HERE, "pairs_64x48.npz"), **out)

    # A small candidate grid (2x2x1 renders x 3x1x1 warps) with its full rating table and winner.
    wl = sy.workload(W, H, 4, 3, seed=11, bottom_up=True)
    ratings, idx, best = oc.search_grid(wl["render_stack"], wl["warp_stack"])
    rn, idxn, bestn = onp.search_grid(wl["render_stack"], wl["warp_stack"])
    assert idx == idxn and np.allclose(ratings, rn, atol=1e-6)
    with oc.rounded():
        ratings_r, idx_r, best_r = oc.search_grid(wl["render_stack"], wl["warp_stack"])
    np.savez_compressed(os.path.join(HERE, "grid_64x48.npz"), render_stack=wl["render_stack"],
                        warp_stack=wl["warp_stack"], ratings=ratings, best_index=np.int64(idx),
                        best_score=np.float32(best), ratings_rounded=ratings_r, best_index_rounded=np.int64(idx_r),
                        best_score_rounded=np.float32(best_r))

    # Known answers at full frame size (inputs are regenerated from the seed, only the expectations are stored).
    a, b = sy.uniform_pair(640, 480, 1234)
    kat = {
        "uniform_640x480_seed1234_suc_topdown": np.float32(oc.eval_pair(a, b, 0, True, False, 1)),
        "uniform_640x480_seed1234_enmi_topdown": np.float32(oc.eval_pair(a, b, 0, True, False, 0)),
        "uniform_640x480_seed1234_suc_bottomup": np.float32(oc.eval_pair(a, b, 0, True, True, 1)),
        "uniform_640x480_seed1234_suc_topdown_rounded": np.float32(_rounded(lambda: oc.eval_pair(a, b, 0, True, False, 1))),
        "uniform_640x480_seed1234_enmi_topdown_rounded": np.float32(_rounded(lambda: oc.eval_pair(a, b, 0, True, False, 0))),
        "uniform_640x480_seed1234_suc_bottomup_rounded": np.float32(_rounded(lambda: oc.eval_pair(a, b, 0, True, True, 1))),
        "uniform_640x480_seed1234_crc": np.uint64(int(a.astype(np.uint64).sum()) * 1000003 + int(b.astype(np.uint64).sum())),
    }
    np.savez_compressed(os.path.join(HERE, "kat_640x480.npz"), **kat)
    print({k: v for k, v in kat.items()})


if __name__ == "__main__":
    main()
