"""CPU tests of the oracle (oracle/): known answers, integer invariants, C restatement vs numpy twin,
committed golden vectors, and the arg-max rule.  The reference has no tests or golden vectors for this path
(SURVEY.md section 4); the analytic values below are the pins listed in SURVEY.md section 8(c)."""
import numpy as np
import pytest

from conftest import dense_joint
from oracle import binding as oc
from oracle import nmi_oracle_np as onp
from orbslam2_nmi_amd import synthetic as sy


def test_known_answer_uniform_pair_640x480(golden_kat):
    # rng(1234).integers twice -> SUC 0.0200299 (fp32 tree 0.020029902, fp64 0.020029872), ENMI 1.0101162
    a, b = sy.uniform_pair(640, 480, 1234)
    crc = np.uint64(int(a.astype(np.uint64).sum()) * 1000003 + int(b.astype(np.uint64).sum()))
    assert crc == golden_kat["uniform_640x480_seed1234_crc"], "numpy RNG stream changed: regenerate tests/golden"
    s = oc.eval_pair(a, b, render_bottom_up=False)
    assert s == np.float32(0.020029902)
    assert abs(float(s) - 0.0200299) < 5e-8
    assert abs(oc.eval_pair_f64(a, b, render_bottom_up=False) - 0.020029872) < 1e-9
    assert abs(float(oc.eval_pair(a, b, render_bottom_up=False, mode=oc.MODE_ENMI)) - 1.0101162) < 2e-7
    assert s == golden_kat["uniform_640x480_seed1234_suc_topdown"]
    assert oc.eval_pair(a, b, render_bottom_up=True) == golden_kat["uniform_640x480_seed1234_suc_bottomup"]
    # fp32 trees vs fp64: the reference's arithmetic is within 1e-7 of the exact value
    assert abs(float(s) - oc.eval_pair_f64(a, b, render_bottom_up=False)) < 1e-7


def test_known_answers_identical_and_constant():
    a, b = sy.uniform_pair(640, 480, 1234)
    assert oc.eval_pair(a, a, render_bottom_up=False) == np.float32(1.0)           # identical -> SUC 1
    assert oc.eval_pair(a, a, render_bottom_up=False, mode=oc.MODE_ENMI) == np.float32(2.0)
    assert oc.eval_pair(a[::-1].copy(), a, render_bottom_up=True) == np.float32(1.0)  # flip restores identity
    c = np.full_like(a, 255)
    z = np.zeros_like(a)
    assert oc.eval_pair(c, z) == 0.0   # both constant: all three sums are 0 -> guard (NMI.cu:353)
    assert oc.eval_pair(c, a) == 0.0   # constant vs anything: H(A)=0, H(A,B)=H(B)
    assert oc.eval_pair(a, c) == 0.0
    assert oc.eval_pair(z, z, use_bg=False) == 0.0  # every pixel skipped -> empty histograms -> 0


def test_integer_invariants():
    B = sy.scene(160, 120, 3)
    F = sy.camera_frame(B, 4)
    F[:10] = 0
    for bg in (True, False):
        for bu in (True, False):
            j, h1, h2 = oc.joint_hist(B, F, 0, bg, bu)
            assert (j.sum(axis=1) == h1).all() and (j.sum(axis=0) == h2).all()
            if bg:
                assert j.sum() == 160 * 120
            else:
                r = B[::-1] if bu else B
                assert j.sum() == np.count_nonzero((r != 0) & (F != 0))
                assert j[0].sum() == 0 and j[:, 0].sum() == 0
    # the flip is a row reversal of the render (NMI.cu:82)
    j_bu, _, _ = oc.joint_hist(B[::-1].copy(), F, 0, True, True)
    j_td, _, _ = oc.joint_hist(B, F, 0, True, False)
    assert (j_bu == j_td).all()


@pytest.mark.parametrize("seed", [0, 1, 2])
@pytest.mark.parametrize("bg", [True, False])
def test_c_restatement_matches_numpy_twin(seed, bg):
    rng = np.random.default_rng(seed)
    w, h = int(rng.integers(17, 90)), int(rng.integers(9, 70))
    r = rng.integers(0, 256, (h, w), dtype=np.uint8)
    f = np.clip(r.astype(int) + rng.integers(-20, 20, (h, w)), 0, 255).astype(np.uint8)
    for bu in (True, False):
        for shift in (0, 2):
            jc = oc.joint_hist(r, f, shift, bg, bu)
            jn = onp.joint_hist(r, f, shift, bg, bu)
            for x, y in zip(jc, jn):
                assert (x == y).all()
            for mode in (oc.MODE_SUC, oc.MODE_ENMI):
                sc, sums = oc.score_from_hist(*jc, w * h, mode)
                sn, sums_n = onp.score_from_hist(*jn, w * h, mode, return_sums=True)
                assert np.allclose(sums, np.array(sums_n), rtol=0, atol=2e-6)
                assert abs(float(sc) - float(sn)) <= 2e-6 * max(1.0, abs(float(sn)))
                assert oc.eval_pair(r, f, shift, bg, bu, mode) == sc


def test_tree_order_is_the_references_not_a_plain_sum():
    # The stride-halving tree (NMI.cu:276-284) and a left-to-right fp32 sum differ in general; the twin's tree and
    # the C tree must agree bit for bit on values where the order matters.
    rng = np.random.default_rng(5)
    j = rng.integers(0, 40, (256, 256)).astype(np.uint32)
    h1, h2 = j.sum(1, dtype=np.uint32), j.sum(0, dtype=np.uint32)
    n = int(j.sum())
    s_c, sums_c = oc.score_from_hist(j, h1, h2, n)
    terms = onp.bin_terms(j, n)
    a3_tree = onp.tree256(onp.tree256(terms))
    a3_seq = np.float32(0)
    for v in terms.reshape(-1):
        a3_seq = np.float32(a3_seq + v)
    assert abs(float(sums_c[2]) - float(a3_tree)) <= 2e-6
    assert a3_seq != a3_tree  # order matters at fp32
    assert abs(float(a3_seq) - float(a3_tree)) < 2e-3


def test_golden_pairs(golden_pairs):
    g = golden_pairs
    for name in g["names"]:
        r, w = g[f"{name}/render"], g[f"{name}/warped"]
        for bg in (1, 0):
            for bu in (1, 0):
                tag = f"{name}/bg{bg}_bu{bu}"
                j, h1, h2 = oc.joint_hist(r, w, 0, bool(bg), bool(bu))
                assert (j == dense_joint(g, tag)).all()
                assert (h1 == g[f"{tag}/hist_render"]).all() and (h2 == g[f"{tag}/hist_warped"]).all()
                for mode in (0, 1):
                    assert oc.eval_pair(r, w, 0, bool(bg), bool(bu), mode) == g[f"{tag}/score_mode{mode}"]
                assert oc.eval_pair(r, w, 2, bool(bg), bool(bu), 1) == g[f"{tag}/score64_bins64"]
    assert g["identical/bg1_bu1/score_mode1"] == np.float32(1.0)
    assert g["constant/bg1_bu1/score_mode1"] == np.float32(0.0)
    assert g["const_vs_smooth/bg1_bu1/score_mode1"] == np.float32(0.0)


def test_golden_grid(golden_grid):
    g = golden_grid
    ratings, idx, best = oc.search_grid(g["render_stack"], g["warp_stack"], threads=2)
    assert (ratings == g["ratings"]).all()
    assert idx == int(g["best_index"]) and best == g["best_score"]


def test_config1_cpu_plumbing_64_bins():
    # BASELINE.json config 1: one frame vs one pre-rendered 640x480 view, 64-bin joint histogram + NMI on the host.
    wl = sy.workload(640, 480, 1, 1, seed=21)
    r, f = wl["render_stack"][0], wl["warp_stack"][0]
    j, h1, h2 = oc.joint_hist(r, f, shift=2)
    assert j[64:].sum() == 0 and j[:, 64:].sum() == 0 and j.sum() == 640 * 480
    s64 = oc.eval_pair(r, f, shift=2)
    s256 = oc.eval_pair(r, f)
    assert 0 < s64 < 1 and 0 < s256 < 1
    assert abs(float(s64) - float(onp.eval_pair(r, f, shift=2))) < 1e-6


def test_find_max_rule():
    # helperFunctions.cpp:52-101 + Tracking.cc:1952: max starts at 0, strict '>', lowest index among ties
    f32 = np.float32
    assert oc.find_max(np.array([0.1, 0.5, 0.5, 0.2], f32)) == (1, f32(0.5))
    assert oc.find_max(np.array([0.0, -0.0, 0.0], f32))[0] == 0
    assert oc.find_max(np.array([-0.1, 0.0, -0.2, 0.0], f32)) == (1, f32(0.0))   # no positive: first exact zero
    assert oc.find_max(np.array([-0.1, -0.2], f32))[0] == -1                        # reference: empty vector
    assert oc.find_max(np.array([np.nan, 0.3, np.nan], f32)) == (1, f32(0.3))
    assert oc.find_max(np.array([np.nan, np.nan], f32))[0] == -1
    for arr in ([0.1, 0.5, 0.5], [-1.0, 0.0], [0.25, np.nan, 0.25]):
        assert oc.find_max(np.array(arr, f32)) == onp.find_max(np.array(arr, f32))


def test_planted_optimum_small():
    wl = sy.workload(160, 120, 27, 8, seed=5)
    ratings, idx, best = oc.search_grid(wl["render_stack"], wl["warp_stack"], threads=4)
    assert idx == wl["planted"]
    assert best == ratings.reshape(-1)[idx]
