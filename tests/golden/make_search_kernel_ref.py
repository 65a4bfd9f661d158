"""Regenerates tests/golden/search_kernel_ref.npz from the REFERENCE's own object code.

oracle/_ref/ref_search_kernel (oracle/Makefile) is /root/reference/Thirdparty/Localization/nmiSearchKernel.cpp compiled
unmodified behind oracle/ref_search_kernel_driver.cpp.  This script feeds it seeded (counts, steps, best indices, NMI,
number of resizes) tuples and stores its answers: isMiddle(), the counts / steps after each resizeKernel(), the
operator<< text before and after, plus a scripted walk over every other public member.  The binary exists only in the
build container (/root/reference does not travel); the fixture is data and does.

Run from the repository root:  make -C oracle && python tests/golden/make_search_kernel_ref.py
"""
import os
import subprocess
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
EXE = os.path.join(ROOT, "oracle", "_ref", "ref_search_kernel")
N_CASES = 3000


def f32bits(x):
    return int(np.float32(x).view(np.uint32))


def cases(n=N_CASES, seed=20261004):
    rng = np.random.default_rng(seed)
    out = []
    # steps around the two collapse thresholds (0.005 m, 0.001 rad as doubles against float steps, nmiSearchKernel.cpp:120-137)
    # and their doubles, the YAML defaults, and log-uniform values
    trans = [0.005, 0.01, 0.0100001, 0.00999999, 0.02, 0.2, 0.5, 0.004, 0.0051, 0.04]
    rot = [0.001, 0.002, 0.0020001, 0.00199999, 0.004, 0.02, 0.05, 0.0009, 0.0011, 0.008]
    for i in range(n):
        num = [int(rng.choice([1, 1, 2, 3, 3, 3, 4, 5, 7, 9])) for _ in range(6)]
        step = []
        for a in range(6):
            pool = trans if a < 3 else rot
            if rng.random() < 0.6:
                step.append(float(rng.choice(pool)))
            else:
                step.append(float(np.exp(rng.uniform(np.log(2e-4), np.log(2.0)))))
        best = []
        for a in range(6):
            u = rng.random()
            if u < 0.35:
                best.append(num[a] // 2)                       # the middle cell (isMiddle's integer n/2)
            elif u < 0.55:
                best.append(0)
            elif u < 0.75:
                best.append(num[a] - 1)
            elif u < 0.8:
                best.append(-1)                                # never set (constructor value)
            else:
                best.append(int(rng.integers(0, num[a])))
        if i % 7 == 0:
            best = [k // 2 for k in num]                       # all-middle cases
        nmi = float(rng.choice([0.0, 0.28606, 1.0, 2.0, 0.123456789, 1e-7, 12345.678])) if rng.random() < 0.5 else float(rng.random())
        R = int(rng.integers(0, 5))                            # the strategy resizes at most nmi_prop_MAX_ITERATION_COUNT - 1 times
        out.append((num, [f32bits(s) for s in step], best, f32bits(nmi), R))
    return out


def main():
    if not os.path.exists(EXE):
        sys.exit(f"{EXE} missing: run `make -C oracle` in a container that has /root/reference")
    cs = cases()
    text = "".join(" ".join(map(str, num)) + " " + " ".join(f"{s:08x}" for s in step) + " " + " ".join(map(str, best))
                   + f" {nmi:08x} {R}\n" for num, step, best, nmi, R in cs)
    out = subprocess.run([EXE], input=text.encode(), capture_output=True, check=True).stdout
    walk = subprocess.run([EXE, "walk"], capture_output=True, check=True).stdout
    assert out.count(b"\n") == 3 * len(cs)
    np.savez_compressed(os.path.join(HERE, "search_kernel_ref.npz"),
                        inputs=np.frombuffer(text.encode(), np.uint8), outputs=np.frombuffer(out, np.uint8),
                        walk=np.frombuffer(walk, np.uint8))
    print(f"wrote {len(cs)} cases, {len(out)} output bytes")


if __name__ == "__main__":
    main()
