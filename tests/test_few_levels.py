"""GPU parity tests (-m gpu) of the few-levels path (csrc/nmi_fewlevels_kernel.hip, NMI_OPT_CONTENT_PATH): frames and
renders with few distinct intensities are scored from rank images with replicated 32-bit counters.  Whatever kernels
score a search, every rating, every winner must equal the oracle's (rounded term mode, ==), and the device-side
fall-back to the general kernel must be invisible."""
import numpy as np
import pytest

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def nmi():
    if not torch.cuda.is_available():
        pytest.fail("gpu tests need a HIP device")
    import orbslam2_nmi_amd as m
    m.load_library()
    return m


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def quantise(a, levels, lo=0, hi=255):
    """Posterise to `levels` values spread over [lo, hi] (not equally spaced when that does not divide)."""
    lut = np.round(lo + (np.arange(256) * levels // 256) * (hi - lo) / max(levels - 1, 1)).astype(np.uint8)
    return lut[a]


def check(nmi, rs, ws, w, h, path, expect_few, bottom_up=True, bins_limit=None, **kw):
    from oracle import binding as oc
    S, Wn = rs.shape[0], ws.shape[0]
    with nmi.NmiContext(w, h, render_bottom_up=bottom_up, **kw) as ctx:
        ctx.set_option(ctx.OPT_CONTENT_PATH, path)
        if bins_limit is not None:
            ctx.set_option(ctx.OPT_FEWLEVELS_BINS, bins_limit)
        ratings = torch.full((Wn, S), -7.0, dtype=torch.float32, device="cuda")
        idx, best = ctx.search_grid(dev(rs), dev(ws), ratings)
        info = ctx.last_content()
    with oc.rounded():
        ro, io, bo = oc.search_grid(rs, ws, render_bottom_up=bottom_up, threads=16, use_bg=kw.get("use_bg", True), mode=kw.get("mode", 1))
    r = ratings.cpu().numpy()
    assert (r.view(np.uint32) == ro.view(np.uint32)).all(), np.abs(r - ro).max()
    assert (idx, best) == (io, bo)
    if expect_few is not None:
        assert info["few_levels"] == expect_few, info
    return info


@pytest.mark.parametrize("levels", [(2, 2), (4, 4), (16, 16), (3, 29), (32, 32), (40, 50), (64, 64), (256, 8)])
def test_posterised_grids_equal_the_oracle(nmi, levels):
    """nr x nw from 4 to 4096 joint bins: 32, 16 and 8 copies of the counters; 81 candidates at 320x240."""
    from orbslam2_nmi_amd import synthetic as sy
    w, h = 320, 240
    wl = sy.workload(w, h, 9, 9, seed=3)
    rs, ws = quantise(wl["render_stack"], levels[0], 3, 250), quantise(wl["warp_stack"], levels[1], 0, 255)
    info = check(nmi, rs, ws, w, h, 1, True)
    assert info["nr"] <= levels[0] and info["nw"] <= levels[1]


def test_full_size_posterised_config2(nmi):
    """640x480, 27 x 27 candidates posterised to 16 and to 4 levels (the two cliff rows of profiles/r03_b/content_sensitivity.txt)."""
    from orbslam2_nmi_amd import synthetic as sy
    wl = sy.workload(640, 480, 27, 27)
    for lv in (16, 4):
        q = 256 // lv
        rs = (wl["render_stack"] // q * q + q // 2).astype(np.uint8)
        ws = (wl["warp_stack"] // q * q + q // 2).astype(np.uint8)
        check(nmi, rs, ws, 640, 480, 1, True)


@pytest.mark.parametrize("use_bg", [True, False])
@pytest.mark.parametrize("bottom_up", [True, False])
@pytest.mark.parametrize("mode", [0, 1])
def test_switches(nmi, use_bg, bottom_up, mode):
    """Background rule (NMI.cu:85: intensity 0 present in both stacks), row order (NMI.cu:82), both score forms."""
    from orbslam2_nmi_amd import synthetic as sy
    w, h = 160, 128
    wl = sy.workload(w, h, 9, 8, seed=11, bottom_up=bottom_up)
    rs, ws = quantise(wl["render_stack"], 6, 0, 255), quantise(wl["warp_stack"], 5, 0, 200)
    assert (rs == 0).any() and (ws == 0).any()
    check(nmi, rs, ws, w, h, 1, True, bottom_up=bottom_up, use_bg=use_bg, mode=mode)


@pytest.mark.parametrize("bins", [128, 64, 32, 16])
def test_reduced_bin_counts_take_the_path_on_ordinary_content(nmi, bins):
    """Fewer than 256 bins (intensity >> shift): textured content has at most `bins` levels per stack, so 64 bins and
    fewer qualify (64 x 64 = 4096 joint bins); 128 bins fall back on the device.  The background rule must be on for
    shifted bins (with it off, bin 0 mixes skipped and kept intensities): off stays with the general kernel."""
    from oracle import binding as oc
    from orbslam2_nmi_amd import synthetic as sy
    w, h = 320, 240
    wl = sy.workload(w, h, 9, 9, seed=6)
    rs, ws = wl["render_stack"], wl["warp_stack"]
    shift = {128: 1, 64: 2, 32: 3, 16: 4}[bins]
    for use_bg in (True, False):
        with nmi.NmiContext(w, h, bins=bins, use_bg=use_bg) as ctx:
            ctx.set_option(ctx.OPT_CONTENT_PATH, 1)
            ratings = torch.zeros((9, 9), dtype=torch.float32, device="cuda")
            idx, best = ctx.search_grid(dev(rs), dev(ws), ratings)
            info = ctx.last_content()
        with oc.rounded():
            ro, io, bo = oc.search_grid(rs, ws, shift=shift, use_bg=use_bg, threads=16)
        assert (ratings.cpu().numpy().view(np.uint32) == ro.view(np.uint32)).all()
        assert (idx, best) == (io, bo)
        assert info["few_levels"] == (use_bg and bins <= 64), (bins, use_bg, info)


def test_ordinary_content_falls_back_on_the_device(nmi):
    """Forced few-levels path on textured content: the probe says no, the gated general kernel scores the search."""
    from orbslam2_nmi_amd import synthetic as sy
    w, h = 320, 240
    wl = sy.workload(w, h, 9, 9, seed=4)
    info = check(nmi, wl["render_stack"], wl["warp_stack"], w, h, 1, False)
    assert info["nr"] * info["nw"] > 4096
    # the same with the background rule off (another instantiation of the gated kernel; zeros planted in both stacks)
    rs, ws = wl["render_stack"].copy(), wl["warp_stack"].copy()
    rs[:, 10:40, 50:90] = 0
    ws[:, 100:130, 200:260] = 0
    check(nmi, rs, ws, w, h, 1, False, use_bg=False)
    # a lower limit (NMI_OPT_FEWLEVELS_BINS): 33 x 64 levels fall back at 2048, pass at the default 4096
    rs, ws = quantise(wl["render_stack"], 33), quantise(wl["warp_stack"], 64)
    info = check(nmi, rs, ws, w, h, 1, None, bins_limit=2048)
    assert info["few_levels"] == (info["nr"] * info["nw"] <= 2048)
    info = check(nmi, rs, ws, w, h, 1, None)
    assert info["few_levels"] == (info["nr"] * info["nw"] <= 4096)


def test_automatic_mode_follows_the_content(nmi):
    """Default options: every search by the general kernel is a content probe as well (its last workgroup posts the bins its
    candidates' marginals held), so posterised content moves the searches to the few-levels path after ONE general search,
    and ordinary content moves them back after one (that search itself falls back on the device)."""
    from oracle import binding as oc
    from orbslam2_nmi_amd import synthetic as sy
    w, h = 320, 240
    wl = sy.workload(w, h, 9, 9, seed=5)
    nat = (wl["render_stack"], wl["warp_stack"])
    pos = (quantise(nat[0], 8), quantise(nat[1], 8))
    with oc.rounded():
        want = {id(nat): oc.search_grid(*nat, threads=16), id(pos): oc.search_grid(*pos, threads=16)}
    seq = [(pos, False), (pos, True), (pos, True), (nat, False), (nat, False), (pos, False), (pos, True), (nat, False), (pos, False), (pos, True)]
    with nmi.NmiContext(w, h) as ctx:
        ratings = torch.zeros((9, 9), dtype=torch.float32, device="cuda")
        d = {id(nat): (dev(nat[0]), dev(nat[1])), id(pos): (dev(pos[0]), dev(pos[1]))}
        for k, (stacks, few) in enumerate(seq):
            idx, best = ctx.search_grid(*d[id(stacks)], ratings)
            ro, io, bo = want[id(stacks)]
            assert (ratings.cpu().numpy().view(np.uint32) == ro.view(np.uint32)).all(), k
            assert (idx, best) == (io, bo), k
            if few is not None:
                assert ctx.last_content()["few_levels"] == few, (k, ctx.last_content())
        # option 0 switches the path off whatever the hint says
        ctx.set_option(ctx.OPT_CONTENT_PATH, 0)
        idx, best = ctx.search_grid(*d[id(pos)], ratings)
        assert (idx, best) == want[id(pos)][1:] and not ctx.last_content()["few_levels"]


def test_non_blocking_shards_and_growing_grids(nmi):
    """Keys of non-blocking launches (no host wait between them) on a context whose grid grows (rank buffers are
    re-allocated), few-levels and ordinary content interleaved."""
    from oracle import binding as oc
    from orbslam2_nmi_amd import synthetic as sy
    w, h = 160, 128
    jobs = []
    for n, (S, Wn, lv) in enumerate([(9, 9, 4), (27, 9, 0), (27, 27, 12), (10, 8, 3), (27, 27, 0)]):
        wl = sy.workload(w, h, S, Wn, seed=20 + n)
        rs, ws = wl["render_stack"], wl["warp_stack"]
        if lv:
            rs, ws = quantise(rs, lv), quantise(ws, lv + 1)
        jobs.append((rs, ws))
    with nmi.NmiContext(w, h) as ctx:
        ctx.set_option(ctx.OPT_CONTENT_PATH, 1)
        keys = [torch.zeros(1, dtype=torch.int64, device="cuda") for _ in jobs]
        tabs = [torch.zeros((ws.shape[0], rs.shape[0]), dtype=torch.float32, device="cuda") for rs, ws in jobs]
        devs = [(dev(rs), dev(ws)) for rs, ws in jobs]
        for (rs, ws), k, t in zip(devs, keys, tabs):
            ctx.search_grid_shard(rs, 0, rs.shape[0], ws, ratings=t, key_out=k, blocking=False)
        ctx.synchronize()
        for (rs, ws), k, t in zip(jobs, keys, tabs):
            with oc.rounded():
                ro, io, bo = oc.search_grid(rs, ws, threads=16)
            assert (t.cpu().numpy().view(np.uint32) == ro.view(np.uint32)).all()
            assert nmi.key_unpack(int(k.cpu().numpy().view(np.uint64)[0])) == (io, bo)


def test_constant_and_two_level_images(nmi):
    """Degenerate joints: constant stacks (1 x 1 bin, score 0 by the all-zero guard, NMI.cu:342-362: first cell wins),
    constant render over a two-level frame, identical two-level pairs (score 1)."""
    w, h = 64, 48
    rng = np.random.default_rng(9)
    two = (rng.integers(0, 2, (9, h, w)) * 200 + 17).astype(np.uint8)
    const = np.full((9, h, w), 255, np.uint8)
    with nmi.NmiContext(w, h, render_bottom_up=False) as ctx:
        ctx.set_option(ctx.OPT_CONTENT_PATH, 1)
        t = torch.zeros((9, 9), dtype=torch.float32, device="cuda")
        assert ctx.search_grid(dev(const), dev(const), t) == (0, 0.0) and (t == 0).all() and ctx.last_content()["few_levels"]
        assert ctx.search_grid(dev(const), dev(two), t) == (0, 0.0) and (t == 0).all()
        idx, best = ctx.search_grid(dev(two), dev(two), t)
        assert best == 1.0 and idx == 0 and (t.cpu().numpy().diagonal() == 1.0).all() and ctx.last_content()["few_levels"]
