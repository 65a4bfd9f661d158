"""GPU tests (-m gpu) of the pixel-range kernel for mid-size grids (csrc/nmi_pix_kernel.hip): P workgroups per candidate,
each adding a range of the pair's pixels into a packed joint histogram of its own, merged at the candidate's owner.
Everything is compared with the CPU oracle in its rounded term mode with == (tests/test_gpu_parity.py explains the bar).
Why these grids: the live strategy's collapsed-axis levels (src/Tracking.cc:2014-2043, nmiSearchKernel.cpp:124-141) and a
rank's share of a sharded 729-candidate grid have 27 ... 128 candidates.  The forced forms (1, P) also run through the whole
pair / small-grid suite of tests/test_gpu_parity.py (split_mode fixture)."""
import time

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


def oracle_grid(wl, **kw):
    from oracle import binding as oc
    with oc.rounded():
        return oc.search_grid(wl["render_stack"], wl["warp_stack"], render_bottom_up=wl["bottom_up"], threads=16, **kw)


@pytest.mark.parametrize("S,Wn,ranges", [(9, 9, 3), (9, 4, 3), (27, 4, 2), (16, 8, 2), (11, 3, 3), (85, 1, 3), (1, 128, 2)])
def test_mid_size_grids_take_the_pixel_range_kernel_and_equal_the_oracle(nmi, S, Wn, ranges):
    from orbslam2_nmi_amd import synthetic as sy
    w, h = 160, 128
    wl = sy.workload(w, h, S, Wn, seed=S * 131 + Wn)
    ro, io, bo = oracle_grid(wl)
    with nmi.NmiContext(w, h, render_bottom_up=wl["bottom_up"]) as ctx:
        cus = ctx.info()["compute_units"]
        t = torch.full((Wn, S), -3.0, device="cuda")
        got = ctx.search_grid(dev(wl["render_stack"]), dev(wl["warp_stack"]), t)
        st = ctx.pix_status()
    if cus == 256:
        assert st["last_launch_ranges"] == ranges, st
    assert st["healed"] == 0
    assert got == (io, bo)
    assert (t.cpu().numpy().view(np.uint32) == ro.view(np.uint32)).all()


@pytest.mark.parametrize("use_bg,bins,mode,bottom_up", [(True, 256, 1, True), (False, 256, 1, False), (True, 64, 0, True), (True, 256, 0, False), (False, 32, 1, True)])
def test_switches(nmi, use_bg, bins, mode, bottom_up):
    """Background rule, bins, ENMI / SUC, render orientation (BG off below 256 bins has no optimistic path: that grid is
    scored by nmi_grid_kernel, and the result is the same)."""
    from oracle import binding as oc
    from orbslam2_nmi_amd import synthetic as sy
    w, h = 176, 96
    wl = sy.workload(w, h, 9, 5, seed=bins + mode)
    rs, ws = wl["render_stack"].copy(), wl["warp_stack"].copy()
    rs[:, :17, :40] = 0   # zeros in both images: the background rule has something to skip
    ws[:, 9:30, 20:90] = 0
    with oc.rounded():
        ro, io, bo = oc.search_grid(rs, ws, render_bottom_up=bottom_up, threads=16, use_bg=use_bg, mode=mode, shift={256: 0, 64: 2, 32: 3}[bins])
    with nmi.NmiContext(w, h, render_bottom_up=bottom_up, use_bg=use_bg, mode=mode, bins=bins) as ctx:
        t = torch.zeros((5, 9), device="cuda")
        got = ctx.search_grid(dev(rs), dev(ws), t)
        if use_bg or bins == 256:
            assert ctx.pix_status()["last_launch_ranges"] in (3, 0)  # (0: fewer compute units than 45 x 2)
    assert got == (io, bo)
    assert (t.cpu().numpy().view(np.uint32) == ro.view(np.uint32)).all()


def test_counter_wraps_in_helpers_owner_and_merge(nmi):
    """640x480, two intensities per image in a fine pattern (no flat chunks): bins of 76,800 hits.  With 2 ranges each
    workgroup stays below 65,536 per bin and only the MERGE wraps; with the whole pair in one bin pattern shifted, helpers
    and owner wrap on their own.  Either way the detector (sum of decoded counters != W*H) sends the candidate to the exact
    path and the scores equal the oracle's."""
    from oracle import binding as oc
    w, h = 640, 480
    yy, xx = np.mgrid[0:h, 0:w]
    a = np.where((xx + yy) % 2 == 0, 10, 200).astype(np.uint8)
    b = np.where((xx // 2 + yy) % 2 == 0, 30, 90).astype(np.uint8)
    c = np.where(xx % 7 == 0, 30, 90).astype(np.uint8)             # one bin pair of ~130,000 hits with `a`-like partners
    rs = np.stack([a, np.where(xx % 3 == 0, 10, 200).astype(np.uint8)] * 16)[:32]
    ws = np.stack([b, c])
    with oc.rounded():
        ro, io, bo = oc.search_grid(rs, ws, render_bottom_up=False, threads=16)
    for ranges in (2, 3, 4):
        with nmi.NmiContext(w, h, render_bottom_up=False) as ctx:
            ctx.set_option(ctx.OPT_SPLIT, 1)
            ctx.set_option(ctx.OPT_SPLIT_PIXELS, ranges)
            t = torch.zeros((2, 32), device="cuda")
            got = ctx.search_grid(dev(rs), dev(ws), t)
            assert ctx.pix_status() == {"last_launch_ranges": ranges, "healed": 0}
        assert got == (io, bo), ranges
        assert (t.cpu().numpy().view(np.uint32) == ro.view(np.uint32)).all(), ranges


def test_flat_regions_travel_as_side_counters(nmi):
    """Large flat regions (render background 255 over a saturated sky / the frame's border 0) are folded into 32-bit side
    counters per workgroup (fold_flat_chunk, nmi_kernels.hip); a helper's side counters travel in its block's header and
    are merged into the owner's -- including the case where the owner's eight are taken."""
    from oracle import binding as oc
    rng = np.random.default_rng(11)
    w, h = 640, 480
    S, Wn = 12, 4
    rs = rng.integers(0, 256, (S, h, w), dtype=np.uint8)
    ws = rng.integers(0, 256, (Wn, h, w), dtype=np.uint8)
    for s in range(S):  # bands of distinct flat pairs down the image: every pixel range meets several
        for band in range(12):
            rs[s, band * 40:band * 40 + 25, :] = 255 - (band % 11) * (s % 3 + 1)
    for v in range(Wn):
        for band in range(12):
            ws[v, band * 40:band * 40 + 25, :] = (band * 17 + v) % 256
    with oc.rounded():
        ro, io, bo = oc.search_grid(rs, ws, render_bottom_up=True, threads=16)
    for ranges in (2, 4):
        with nmi.NmiContext(w, h) as ctx:
            ctx.set_option(ctx.OPT_SPLIT, 1)
            ctx.set_option(ctx.OPT_SPLIT_PIXELS, ranges)
            t = torch.zeros((Wn, S), device="cuda")
            got = ctx.search_grid(dev(rs), dev(ws), t)
        assert got == (io, bo)
        assert (t.cpu().numpy().view(np.uint32) == ro.view(np.uint32)).all()


def test_a_missing_helper_is_healed_inside_the_launch(nmi):
    """Helpers never wait and are dispatched before the owners, so an owner's wait cannot deadlock; should a helper's flag not
    arrive all the same (here: helper 1 of every candidate is told to withhold it), the owner gives up after 2 ms, scores the
    candidate alone on the exact path and counts the event.  The call's result is the oracle's; nothing is redone by the host."""
    from orbslam2_nmi_amd import synthetic as sy
    w, h = 160, 128
    wl = sy.workload(w, h, 9, 5, seed=3)
    ro, io, bo = oracle_grid(wl)
    rs, ws = dev(wl["render_stack"]), dev(wl["warp_stack"])
    with nmi.NmiContext(w, h, render_bottom_up=wl["bottom_up"]) as ctx:
        ctx.set_option(ctx.OPT_SPLIT, 1)
        ctx.set_option(ctx.OPT_SPLIT_PIXELS, 3)
        assert ctx.search_grid(rs, ws) == (io, bo)
        ctx.set_option(ctx.OPT_PHASE_MASK, 3 | 512)
        t = torch.zeros((5, 9), device="cuda")
        t0 = time.perf_counter()
        assert ctx.search_grid(rs, ws, t) == (io, bo)
        assert time.perf_counter() - t0 < 0.05
        assert (t.cpu().numpy().view(np.uint32) == ro.view(np.uint32)).all()
        assert ctx.pix_status() == {"last_launch_ranges": 3, "healed": 45}
        ctx.set_option(ctx.OPT_PHASE_MASK, 3)
        assert ctx.search_grid(rs, ws) == (io, bo)          # the stale blocks of the failed launch carry an old tag
        assert ctx.pix_status()["healed"] == 45


def test_enqueue_only_calls_and_repeated_launches(nmi):
    """No residence condition: calls that only enqueue (device key, no host wait) use the kernel too; 50 launches back to back
    on one context reuse the same blocks under fresh tags."""
    from orbslam2_nmi_amd import synthetic as sy
    w, h = 160, 128
    wl = sy.workload(w, h, 9, 9, seed=8)
    ro, io, bo = oracle_grid(wl)
    rs, ws = dev(wl["render_stack"]), dev(wl["warp_stack"])
    with nmi.NmiContext(w, h, render_bottom_up=wl["bottom_up"]) as ctx:
        keys = torch.zeros(50, dtype=torch.int64, device="cuda")
        for i in range(50):
            ctx.search_grid_shard(rs, 0, 9, ws, key_out=keys[i:i + 1], blocking=False)
        ctx.synchronize()
        assert ctx.pix_status()["last_launch_ranges"] in (3, 0)
        from orbslam2_nmi_amd import capi
        for k in keys.cpu().numpy():
            assert capi.key_unpack(int(k)) == (io, bo)


def test_captured_level_with_a_mid_size_grid(nmi):
    """nmi_level_* (one search level as a captured HIP graph): a 9 x 9 level's search is the pixel-range kernel inside the graph.
    A replayed graph's arguments are frozen, so the hand-off tag comes from the epoch captured + the replay count the level's
    prep kernel keeps in device memory: eight replays with different views and warps, every rating table == the oracle's on the
    stacks that replay produced (a stale block taken for a fresh one would show at once), for a point cloud and a textured mesh."""
    from oracle import binding as oc
    from orbslam2_nmi_amd import capi, hostapi as H, synthetic as sy
    import sys, os
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from test_render import plane_cloud
    w, h = 320, 240
    xyz, red, rp = plane_cloud(w, h, density=2.0)
    Twc = np.eye(4, dtype=np.float32)
    Twc[:3, 1] = [0, -1, 0]
    pos, look, up = Twc[:3, 3], Twc[:3, 3] + Twc[:3, 2], Twc[:3, 1]
    cells = [(sx, sy_, 0) for sy_ in range(3) for sx in range(3)]
    K = sy.intrinsics(w, h)
    with nmi.NmiContext(w, h) as ctx:
        dx, dr = torch.from_numpy(xyz).cuda(), torch.from_numpy(red).cuda()
        frame = torch.flip(ctx.render_points(dx, torch.sqrt(dr), capi.render_mvp(rp, pos, look, up, (0.05, 0, 0))[None], 3.0)[0], dims=[0]).contiguous()
        with nmi.NmiLevel(ctx, dx, dr, frame, 9, 9, 3.0) as lv:
            for rep in range(8):
                g = H.SearchKernel.make([3, 3, 1, 3, 3, 1], [s / (1 + 0.3 * rep) for s in (0.2, 0.2, 0.5, 0.02, 0.02, 0.05)])
                mvps = np.stack([capi.render_mvp(rp, pos, look, up, H.calculate_translation(Twc, g, *c)) for c in cells])
                Ms = capi.warp_homographies(K, (3, 3, 1), tuple(g.step[3:6]))
                win = lv.run(mvps, Ms)
                rs, ws, table = lv.outputs()
                with oc.rounded():
                    ro, io, bo = oc.search_grid(rs, ws, render_bottom_up=True, threads=16)
                assert (table.view(np.uint32) == ro.view(np.uint32)).all(), rep
                assert win == (io, bo), rep
        assert ctx.pix_status()["healed"] == 0
