"""CPU tests of the host-side driver (include/nmi_host.h): NmiSearchKernel arithmetic, the arg-max rule, the pose
update and the coarse-to-fine strategy, each against a table or an independent Python model written from the
reference's behaviour (nmiSearchKernel.cpp:99-141,183-195; helperFunctions.cpp:50-103; rendering.hpp:668-694;
Tracking.cc:1987-2179,2374-2419).  The reference has no tests of its own for these."""
import ctypes as C
import math
import os
import re

import numpy as np
import pytest

from conftest import ROOT
from orbslam2_nmi_amd import build as nmi_build
from orbslam2_nmi_amd import hostapi as H

f32 = np.float32


@pytest.fixture(scope="module", autouse=True)
def _built():
    nmi_build.build()


def test_header_symbols_exported():
    text = re.sub(r"/\*.*?\*/", "", open(os.path.join(ROOT, "include", "nmi_host.h")).read(), flags=re.S)
    declared = sorted(set(re.findall(r"\b(nmi_[a-z0-9_]+)\s*\(", text)) - {"nmi_eval_grid_fn"})
    assert declared == sorted(H.EXPORTED_SYMBOLS)
    raw = C.CDLL(nmi_build.LIB)
    for name in declared:
        assert hasattr(raw, name), name


def test_defaults_equal_reference_macros():
    p = H.properties_default()
    assert p.max_iteration_count == 4 and p.reloc_frequency == 2           # allProperties.hpp:27,31
    assert p.step_factor == 0.5 and p.use_bg == 1                           # :33,38
    assert p.min_kernel_rotation == 0.001 and p.min_kernel_translation == 0.005  # :49-50
    k = H.SearchKernel()
    H._lib().nmi_sk_init(C.byref(k))
    assert list(k.num) == [-1] * 6 and list(k.best) == [-1] * 6 and list(k.step) == [-1.0] * 6 and k.nmi == 0


def test_is_middle_uses_integer_half():
    k = H.SearchKernel.make([3, 3, 3, 3, 3, 3], [1] * 6, best=[1, 1, 1, 1, 1, 1])
    assert H.is_middle(k)
    k.best[4] = 2
    assert not H.is_middle(k)
    k = H.SearchKernel.make([4, 1, 5, 2, 1, 3], [1] * 6, best=[2, 0, 2, 1, 0, 1])  # n/2 with integer division
    assert H.is_middle(k)


RESIZE_CASES = [
    # (num, step, best) -> (num', step')
    (([3, 3, 3, 3, 3, 3], [0.2, 0.2, 0.5, 0.02, 0.02, 0.05], [1, 1, 1, 1, 1, 1]),
     ([3, 3, 3, 3, 3, 3], [0.1, 0.1, 0.25, 0.01, 0.01, 0.025])),
    # best on the border of a multi-cell axis keeps its step; single-cell axes always halve
    (([3, 3, 1, 3, 1, 3], [0.2, 0.2, 0.5, 0.02, 0.02, 0.05], [0, 2, 0, 2, 0, 1]),
     ([3, 3, 1, 3, 1, 3], [0.2, 0.2, 0.25, 0.02, 0.01, 0.025])),
    # steps falling under 0.005 m / 0.001 rad collapse the axis to one cell (also when it was on the border before)
    (([3, 3, 3, 3, 3, 3], [0.008, 0.004, 0.02, 0.0015, 0.0009, 0.004], [1, 0, 1, 1, 2, 1]),
     ([1, 1, 3, 1, 1, 3], [0.004, 0.004, 0.01, 0.00075, 0.0009, 0.002])),
]


@pytest.mark.parametrize("case", RESIZE_CASES)
def test_resize_kernel_table(case):
    (num, step, best), (num2, step2) = case
    k = H.resize(H.SearchKernel.make(num, step, best=best, nmi=0.3))
    assert list(k.num) == num2
    assert np.allclose(list(k.step), np.array(step2, f32), rtol=1e-7, atol=0)
    assert list(k.best) == best and k.nmi == f32(0.3)  # resize never touches the best indices or the score


def test_resize_threshold_compares_float_against_double():
    # 0.005f is slightly below 0.005 (double): a float step of exactly 0.005f collapses the axis, as in the reference
    k = H.SearchKernel.make([3] * 6, [0.01, 1, 1, 1, 1, 1], best=[1] * 6)
    H.resize(k)
    assert k.step[0] == f32(0.005) and (float(f32(0.005)) < 0.005) and k.num[0] == 1


def test_format_matches_reference_stream_format():
    k = H.SearchKernel.make([3, 3, 3, 3, 3, 3], [0.2, 0.2, 0.5, 0.02, 0.02, 0.05], best=[1, 0, 2, 1, 1, 1], nmi=0.28606)
    exp = ("sX:  1/3: 0.20000;\t sY:  0/3: 0.20000;\t sZ:  2/3: 0.50000;\t rX:  1/3: 0.02000;\t rY:  1/3: 0.02000;"
           "\t rZ:  1/3: 0.05000;\t NMI: 0.28606")
    assert H.fmt(k) == exp
    k2 = H.SearchKernel.make([5, 5, 5, 3, 3, 3], [0.2] * 6)  # best = -1 prints with width 2
    assert H.fmt(k2).startswith("sX: -1/5: 0.20000;")


def test_linear_index_is_the_find_max_scan_order():
    k = H.SearchKernel.make([2, 3, 4, 3, 2, 5], [1] * 6)
    assert H.candidates(k) == 2 * 3 * 4 * 3 * 2 * 5
    seen = []
    for wz in range(5):
        for wy in range(2):
            for wx in range(3):
                for sz in range(4):
                    for sy in range(3):
                        for sx in range(2):
                            seen.append(H.linear_index(k, [sx, sy, sz, wx, wy, wz]))
    assert seen == list(range(H.candidates(k)))
    for lin in (0, 1, 17, 719):
        H.set_best_from_index(k, lin, 0.5)
        assert H.linear_index(k, list(k.best)) == lin and k.nmi == 0.5
    assert H.set_best_from_index(k, 720, 0.1) != 0 and H.linear_index(k, [2, 0, 0, 0, 0, 0]) == -1
    # the GPU rating layout: index = w*S + s with s = (sz*nSy+sy)*nSx+sx, w = (wz*nWy+wy)*nWx+wx
    S = 2 * 3 * 4
    assert H.linear_index(k, [1, 2, 3, 2, 1, 4]) == ((4 * 2 + 1) * 3 + 2) * S + ((3 * 3 + 2) * 2 + 1)


def test_find_max_elements_ties_and_degenerate():
    assert H.find_max_elements([0.1, 0.5, 0.2, 0.5, 0.5]) == ([1, 3, 4], 3, f32(0.5))
    assert H.find_max_elements([-0.1, 0.0, -0.3, 0.0]) == ([1, 3], 2, f32(0))
    assert H.find_max_elements([-0.1, -0.2]) == ([], 0, f32(0))        # reference: ExtremElements[0] on an empty vector
    assert H.find_max_elements([float("nan"), 0.25])[0] == [1]
    assert H.find_max_elements([0.5, 0.5, 0.5], cap=2)[:2] == ([0, 1], 3)


def rot(axis, a):
    c, s = math.cos(a), math.sin(a)
    return {"x": np.array([[1, 0, 0], [0, c, -s], [0, s, c]]), "y": np.array([[c, 0, s], [0, 1, 0], [-s, 0, c]]),
            "z": np.array([[c, -s, 0], [s, c, 0], [0, 0, 1]])}[axis]


def random_pose(rng):
    T = np.eye(4)
    T[:3, :3] = rot("z", rng.uniform(-3, 3)) @ rot("y", rng.uniform(-1, 1)) @ rot("x", rng.uniform(-3, 3))
    T[:3, 3] = rng.uniform(-50, 50, 3)
    return T


def model_translation(Twc, num, step, s):
    """rendering.hpp:668-694 with the camera of ioData.cpp:177-197, in float64."""
    up, view = Twc[:3, 1], Twc[:3, 2]
    dy = up / np.linalg.norm(up)
    dz = -view / np.linalg.norm(view)
    dx = np.cross(dy, dz)  # dy rotated by -90 degrees about dz
    return sum((s[a] - (num[a] - 1) / 2.0) * step[a] * d for a, d in enumerate((dx, dy, dz)))


def model_relocalization(Twc, k):
    """Tracking.cc:2374-2419 in float64."""
    num, step, best = list(k.num), [float(x) for x in k.step], list(k.best)
    ang = [(best[3 + a] - num[3 + a] // 2) * step[3 + a] for a in range(3)]
    N = np.eye(4)
    N[:3, :3] = rot("z", ang[2]) @ rot("y", ang[1]) @ rot("x", ang[0])
    out = Twc @ N
    out[:3, 3] += model_translation(Twc, num, step, best[:3])
    return out


def test_pose_update_matches_model():
    rng = np.random.default_rng(0)
    for _ in range(50):
        Twc = random_pose(rng)
        num = [int(rng.integers(1, 6)) for _ in range(6)]
        step = [rng.uniform(0.01, 0.5) for _ in range(3)] + [rng.uniform(0.001, 0.05) for _ in range(3)]
        best = [int(rng.integers(0, n)) for n in num]
        k = H.SearchKernel.make(num, step, best=best)
        t = H.calculate_translation(Twc, k, *best[:3])
        assert np.allclose(t, model_translation(Twc, num, [float(f32(s)) for s in step], best[:3]), atol=2e-5)
        got = H.calculate_relocalization(Twc, k)
        assert np.allclose(got, model_relocalization(Twc.astype(f32).astype(float), k), atol=5e-5)
        assert np.allclose(H.mat4_inverse(Twc) @ Twc.astype(f32), np.eye(4), atol=1e-4)
    # centre cell of an odd grid = no motion
    k = H.SearchKernel.make([3] * 6, [0.2] * 6, best=[1] * 6)
    T = random_pose(rng)
    assert np.allclose(H.calculate_relocalization(T, k), T.astype(f32), atol=1e-5)
    # even counts: rotations use n/2 (integer), translations use (n-1)/2 (float) -- the reference's asymmetry
    k = H.SearchKernel.make([4] * 6, [0.2] * 6, best=[2] * 6)
    got = H.calculate_relocalization(np.eye(4), k)
    assert np.allclose(got[:3, :3], np.eye(3), atol=1e-6)            # 2 - 4/2 = 0 rotation
    assert np.allclose(np.abs(got[:3, 3]), 0.1, atol=1e-6)            # (2 - 1.5) * 0.2 along each camera axis


# ---- strategy state machine vs an independent Python model -------------------------------------------------

def py_resize(k, props):
    k = dict(num=list(k["num"]), step=[f32(s) for s in k["step"]], best=list(k["best"]), nmi=k["nmi"])
    for a in range(6):
        border = (k["best"][a] == k["num"][a] - 1 or k["best"][a] == 0) and k["num"][a] > 1
        if not border:
            k["step"][a] = f32(k["step"][a] * f32(props.step_factor))
    for a in range(6):
        lim = props.min_kernel_translation if a < 3 else props.min_kernel_rotation
        if float(k["step"][a]) < lim:
            k["num"][a] = 1
    return k


def py_strategy(Tcw, init, evalf, dist, rotn, not_init, thr, props):
    """Tracking.cc:1987-2179 re-derived: returns (Tcw, relocalized, failed, iterations, stop, per-iteration kernels)."""
    blank = dict(num=[-1] * 6, step=[f32(-1)] * 6, best=[-1] * 6, nmi=f32(0))
    cur, last = dict(blank), dict(blank)
    if dist[0] > 0:
        steps = [f32(f32(d) * 0.02) for d in list(dist) + list(rotn)]
        cur = dict(num=[1 if float(steps[a]) < (props.min_kernel_translation if a < 3 else props.min_kernel_rotation)
                        else init["num"][a] for a in range(6)], step=steps, best=[-1] * 6, nmi=f32(0))
    elif not_init:
        cur = dict(num=[5, 5, 5] + init["num"][3:], step=list(init["step"]), best=[-1] * 6, nmi=f32(0))
    else:
        cur = dict(num=list(init["num"]), step=list(init["step"]), best=[-1] * 6, nmi=f32(0))
    Tcw = np.asarray(Tcw, f32)
    save, save_last = Tcw.copy(), Tcw.copy()
    i = under = 0
    stop, hist = 0, []
    while True:
        i += 1
        if i > props.max_iteration_count:
            break
        Twc = np.linalg.inv(Tcw.astype(float)).astype(f32)
        idx, sc = evalf(cur, Twc)
        b = []
        rem = idx
        for a in range(6):
            b.append(rem % cur["num"][a])
            rem //= cur["num"][a]
        cur = dict(cur, best=b, nmi=f32(sc))
        k = H.SearchKernel.make(cur["num"], cur["step"], best=b)
        Tcw = np.linalg.inv(model_relocalization(Twc.astype(float), k)).astype(f32)
        hist.append(dict(cur))
        if i > 1 and all(cur["best"][a] == cur["num"][a] // 2 for a in range(6)):
            stop = 1
            break
        if i > 1:
            with np.errstate(divide="ignore", invalid="ignore"):
                ratio = float(f32(cur["nmi"]) / f32(last["nmi"]))
            if ratio < 1.001:
                if under > 0:
                    stop = 2
                    break
                under += 1
            else:
                under = 0
        last = dict(cur)
        cur = py_resize(cur, props)
        save_last = Tcw.copy()
    reverted = bool(cur["nmi"] < last["nmi"])
    if reverted:
        Tcw = save_last
    d = math.sqrt(sum(float(f32(x)) ** 2 for x in dist))
    t = float(f32(thr)) if d < 5 else max(float(f32(thr)) * (5 / d), float(f32(thr) / f32(2)))
    reloc, failed = len(hist) > 0, False
    if float(cur["nmi"]) < t:
        Tcw, reloc, failed = save, False, True
    return Tcw, reloc, failed, len(hist), stop, hist, reverted, cur, last


def make_eval(target_Twc, sharp=4.0, flat_after=None):
    """Synthetic scorer: every candidate of the grid is scored by its pose distance to a hidden target."""
    calls = []

    def evalf(kern, Twc):
        if isinstance(kern, dict):
            num, step = kern["num"], kern["step"]
        else:
            num, step = list(kern.num), list(kern.step)
        k = H.SearchKernel.make(num, step)
        n = int(np.prod(num))
        scores = np.zeros(n, f32)
        for lin in range(n):
            H.set_best_from_index(k, lin, 0)
            P = H.calculate_relocalization(Twc, k).astype(float)
            dt = np.linalg.norm(P[:3, 3] - target_Twc[:3, 3])
            dr = np.linalg.norm(P[:3, :3] - target_Twc[:3, :3])
            scores[lin] = f32(0.9 * math.exp(-sharp * (dt * dt + 25 * dr * dr)))
        if flat_after is not None and len(calls) >= flat_after:
            scores[:] = scores.max()  # no more gain: exercises the 0.1 % rule and the tie-break
        calls.append(n)
        ties, cnt, mx = H.find_max_elements(scores, cap=1)
        return ties[0], mx

    return evalf, calls


@pytest.mark.parametrize("seed", range(8))
def test_strategy_matches_model(seed):
    rng = np.random.default_rng(100 + seed)
    props = H.properties_default()
    Twc0 = random_pose(rng)
    target = Twc0.copy()
    target[:3, 3] += rng.uniform(-0.3, 0.3, 3)
    target[:3, :3] = target[:3, :3] @ rot("x", rng.uniform(-0.03, 0.03)) @ rot("z", rng.uniform(-0.05, 0.05))
    Tcw0 = np.linalg.inv(Twc0).astype(f32)
    init = dict(num=[3, 3, 3, 3, 3, 3], step=[f32(s) for s in (0.2, 0.2, 0.5, 0.02, 0.02, 0.05)])
    kinit = H.SearchKernel.make(init["num"], init["step"])
    mode = seed % 4
    dist = rotn = (0, 0, 0)
    not_init = False
    flat = None
    thr = 0.1
    if mode == 1:
        dist, rotn = tuple(rng.uniform(0.1, 12, 3)), tuple(rng.uniform(0.01, 2, 3))  # drift-seeded grid, maybe > 5 m
    elif mode == 2:
        not_init, init["num"] = True, [3, 3, 3, 3, 1, 3]
        kinit = H.SearchKernel.make(init["num"], init["step"])
    elif mode == 3:
        flat, thr = 1, 0.95  # stalls after the first level and is rejected by the threshold
    e1, calls1 = make_eval(target, flat_after=flat)
    e2, calls2 = make_eval(target, flat_after=flat)
    out = H.relocalize_with_strategy(Tcw0, kinit, e1, dist, rotn, not_init, thr, props)
    mT, mre, mfail, mit, mstop, mhist, mrev, mcur, mlast = py_strategy(Tcw0, init, e2, dist, rotn, not_init, thr, props)
    assert calls1 == calls2
    assert (out.iterations, out.stop_reason, bool(out.relocalized), bool(out.failed)) == (mit, mstop, mre, mfail)
    # "final NMI below the previous iterate" is decided by a float '<': when the two levels score the same to
    # within the float noise of the two pose pipelines (C++ vs numpy inverse) either outcome is legitimate.
    near_tie = abs(float(mcur["nmi"]) - float(mlast["nmi"])) < 1e-5
    if bool(out.reverted_to_previous) == mrev:
        assert np.allclose(np.array(out.Tcw).reshape(4, 4), mT, atol=2e-4)
    else:
        assert near_tie
    for i, h in enumerate(mhist):
        assert list(out.per_iteration[i].num) == h["num"] and list(out.per_iteration[i].best) == h["best"]
        assert np.allclose(list(out.per_iteration[i].step), h["step"], rtol=1e-6)
        assert abs(out.per_iteration[i].nmi - h["nmi"]) < 2e-5
    assert list(out.kernel.num) == mcur["num"] and np.allclose(list(out.kernel.step), mcur["step"], rtol=1e-6)
    assert list(out.last_kernel.num) == mlast["num"]
    if mode == 2:
        assert list(out.per_iteration[0].num)[:3] == [5, 5, 5]      # Tracking.cc:2057
    if mode == 3:
        assert out.failed and np.allclose(np.array(out.Tcw).reshape(4, 4), Tcw0, atol=0)  # pose restored bit for bit
    if mode == 0:
        assert out.iterations <= 4 and not out.failed
        # the search moved the pose towards the hidden target
        before = np.linalg.norm(Twc0[:3, 3] - target[:3, 3])
        after = np.linalg.norm(np.linalg.inv(np.array(out.Tcw).reshape(4, 4).astype(float))[:3, 3] - target[:3, 3])
        assert after <= before + 1e-6


def test_strategy_threshold_scaling_beyond_five_metres():
    props = H.properties_default()
    kinit = H.SearchKernel.make([1] * 6, [0.2] * 6)
    ev = lambda k, T: (0, 0.06)
    for dist, exp_thr in (((3, 0, 0), 0.1), ((8, 0, 0), 0.1 * 5 / 8), ((30, 40, 0), 0.05)):
        out = H.relocalize_with_strategy(np.eye(4), kinit, ev, dist, (0, 0, 0), False, 0.1, props)
        assert abs(out.nmi_threshold_used - exp_thr) < 1e-6
        assert bool(out.failed) == (0.06 < exp_thr)
