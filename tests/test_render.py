"""Point-cloud render-stack producer: host matrices vs a float64 model of rendering.hpp:196-202 /
glm::lookAt (CPU), the HIP splat kernel vs the fp32 numpy restatement (GPU, bit-exact), and frame -> device renders ->
search end to end.  Parity with an OpenGL driver is unpinned (oracle/render_oracle_np.py header)."""
import numpy as np
import pytest

from orbslam2_nmi_amd import capi, synthetic as sy
from orbslam2_nmi_amd import build as nmi_build
from oracle import render_oracle_np as ro


@pytest.fixture(scope="module", autouse=True)
def _built():
    nmi_build.build()


def params(w, h, point_size=3.0, zn=5.0, zf=30.0):
    K = sy.intrinsics(w, h)
    return capi.RenderParams(fx=K[0, 0], fy=K[1, 1], cx=K[0, 2], cy=K[1, 2], near_plane=zn, far_plane=zf, point_size=point_size)


def test_mvp_matches_float64_model():
    rng = np.random.default_rng(0)
    rp = params(640, 480)
    for _ in range(20):
        pos, tr = rng.uniform(-20, 20, 3), rng.uniform(-1, 1, 3)
        d = rng.standard_normal(3)
        d /= np.linalg.norm(d)
        up = np.cross(d, rng.standard_normal(3))
        got = capi.render_mvp(rp, pos, pos + d, up, tr).reshape(4, 4).T  # column-major -> conventional
        exp = ro.projection(rp.fx, rp.fy, rp.cx, rp.cy, rp.near_plane, rp.far_plane) @ ro.look_at(pos + tr, pos + d + tr, up)
        assert np.allclose(got, exp, rtol=2e-5, atol=2e-4)
    # a point straight ahead at mid depth lands on the image centre in clip space
    m = capi.render_mvp(rp, (0, 0, 0), (0, 0, 1), (0, -1, 0), (0, 0, 0)).reshape(4, 4).T
    c = m @ np.array([0, 0, 10.0, 1.0])
    assert abs(c[0]) < 1e-6 and abs(c[1]) < 1e-6 and c[3] == pytest.approx(10.0) and -c[3] < c[2] < c[3]


def plane_cloud(w, h, depth=10.0, density=1.6, seed=3):
    """A fronto-parallel textured plane in front of a camera at the origin looking along +z: one point per ~1/density px."""
    B = sy.scene(2 * w, 2 * h, seed)
    rp = params(w, h)
    n_u, n_v = int(2 * w * density / 2), int(2 * h * density / 2)
    u = np.linspace(-w, 2 * w, n_u)            # pixel-ish coordinates over an area larger than the frame
    v = np.linspace(-h, 2 * h, n_v)
    uu, vv = np.meshgrid(u, v)
    X = (uu - rp.cx) / rp.fx * depth
    Y = (vv - rp.cy) / rp.fy * depth
    xyz = np.stack([X, Y, np.full_like(X, depth)], -1).reshape(-1, 3).astype(np.float32)
    ti = np.clip(((uu + w) / 3 * 2).astype(int), 0, 2 * w - 1)
    tj = np.clip(((vv + h) / 3 * 2).astype(int), 0, 2 * h - 1)
    red = (B[tj, ti].astype(np.float32) / np.float32(256.0)).reshape(-1)   # objloader.cpp:261: colour / 256
    return xyz, red, rp


def test_twin_sanity_single_point():
    rp = params(64, 48, point_size=3)
    m = capi.render_mvp(rp, (0, 0, 0), (0, 0, 1), (0, -1, 0), (0, 0, 0))
    img = ro.render_points(np.array([[0, 0, 10.0]], np.float32), np.array([100 / 256], np.float32), m, 64, 48, 3)
    ys, xs = np.nonzero(img != 255)
    assert len(ys) == 9 and img[ys[0], xs[0]] == 100                      # 3x3 sprite, colour round(100/256*255)
    assert xs.min() == 31 and xs.max() == 33 and ys.min() == 23 and ys.max() == 25   # centred on the window centre
    # behind the camera / beyond the far plane / outside the frustum: nothing drawn
    for p in ([0, 0, -10.0], [0, 0, 40.0], [1000.0, 0, 10.0], [0, 0, 2.0]):
        assert (ro.render_points(np.array([p], np.float32), np.array([0.5], np.float32), m, 64, 48, 3) == 255).all()
    # depth test: the nearer point wins regardless of draw order
    two = np.array([[0, 0, 20.0], [0, 0, 10.0]], np.float32)
    for order in ([0, 1], [1, 0]):
        img = ro.render_points(two[order], np.array([0.9, 0.1], np.float32)[order], m, 64, 48, 1)
        assert img[24, 32] == int(0.1 * 255 + 0.5)


@pytest.mark.gpu
@pytest.mark.parametrize("point_size,shape", [(1.0, (160, 120)), (2.0, (160, 120)), (3.0, (160, 120)), (4.4, (160, 120)),
                                              (5.0, (160, 120)), (7.0, (160, 120)),   # 7: the any-size resolve kernel
                                              (3.0, (150, 90)), (1.0, (150, 90))])     # width % 4 != 0: likewise
def test_gpu_splat_bit_exact_vs_twin(point_size, shape):
    torch = pytest.importorskip("torch")
    import orbslam2_nmi_amd as nmi
    w, h = shape
    xyz, red, rp = plane_cloud(w, h)
    rng = np.random.default_rng(5)
    extra = rng.uniform(-30, 30, (5000, 3)).astype(np.float32)          # clutter at random depths incl. behind / too near / too far
    xyz = np.concatenate([xyz, extra])
    red = np.concatenate([red, rng.uniform(-0.2, 1.2, 5000).astype(np.float32)])
    mvps = np.stack([capi.render_mvp(rp, (0, 0, 0), (0, 0, 1), (0, -1, 0), t)
                     for t in ((0, 0, 0), (0.3, -0.2, 0.5), (-1.0, 0.4, -2.0))])
    with nmi.NmiContext(w, h) as ctx:
        got = ctx.render_points(torch.from_numpy(xyz).cuda(), torch.from_numpy(red).cuda(), mvps, point_size).cpu().numpy()
    exp = ro.render_stack(xyz, red, mvps, w, h, point_size)
    assert got.shape == exp.shape == (3, h, w)
    assert (got == exp).all(), f"{(got != exp).sum()} pixels differ"
    assert (exp != 255).mean() > (0.25 if point_size < 2 else 0.6)        # the plane actually covers the view


@pytest.mark.gpu
def test_gpu_splat_block_culling_changes_nothing():
    """The splat kernel skips a view for a block of 256 consecutive points whose bounding box lies beyond one clip plane.
    A dense cloud several times larger than the view, seen from rotated and displaced cameras (oblique clip planes, boxes
    that straddle them, boxes partly behind the camera), must still equal the twin, which tests every point."""
    torch = pytest.importorskip("torch")
    import orbslam2_nmi_amd as nmi
    w, h = 96, 64
    xyz, red, rp = plane_cloud(w, h, density=5.0)
    rng = np.random.default_rng(8)
    wall = xyz[::3].copy()                       # a second, tilted sheet crossing the first and the near plane
    wall[:, 2] = 4.0 + 0.8 * wall[:, 0] + 0.1 * wall[:, 1]
    xyz = np.concatenate([xyz, wall.astype(np.float32)])
    red = np.concatenate([red, red[::3][::-1]])
    views = []
    for k in range(14):
        ang_y, ang_x = rng.uniform(-0.8, 0.8), rng.uniform(-0.5, 0.5)
        d = np.array([np.sin(ang_y) * np.cos(ang_x), np.sin(ang_x), np.cos(ang_y) * np.cos(ang_x)])
        pos = rng.uniform(-3, 3, 3) * np.array([1, 1, 0.5])
        views.append(capi.render_mvp(rp, pos, pos + d, (0, -1, 0), rng.uniform(-0.5, 0.5, 3)))
    views.append(capi.render_mvp(rp, (0, 0, 20.0), (0, 0, 19.0), (0, -1, 0), (0, 0, 0)))   # looking back at the sheet
    mvps = np.stack(views)
    with nmi.NmiContext(w, h) as ctx:
        got = ctx.render_points(torch.from_numpy(xyz).cuda(), torch.from_numpy(red).cuda(), mvps, 2.0).cpu().numpy()
    exp = ro.render_stack(xyz, red, mvps, w, h, 2.0)
    assert (got == exp).all(), f"{(got != exp).sum()} pixels differ"
    covered = (exp != 255).reshape(len(views), -1).mean(1)
    assert (covered > 0.3).sum() >= 8 and (covered < 0.999).any()


@pytest.mark.gpu
def test_gpu_cloud_to_winner_end_to_end():
    """cloud + pose -> device render stack for a 3x3x3 translation grid; frame = the view from a displaced pose; the
    search must pick the cell nearest to the displacement (no OpenGL, no host-side rendering of the stack)."""
    torch = pytest.importorskip("torch")
    import orbslam2_nmi_amd as nmi
    from orbslam2_nmi_amd import hostapi as H
    w, h = 320, 240
    xyz, red, rp = plane_cloud(w, h, density=2.0)
    Twc = np.eye(4, dtype=np.float32)
    Twc[:3, 1] = [0, -1, 0]                       # camera up = -y, view = +z (setupCam reads these columns, ioData.cpp:177-197)
    grid = H.SearchKernel.make([3, 3, 3, 1, 1, 1], [0.2, 0.2, 0.5, 0.02, 0.02, 0.05])
    pos, look, up = Twc[:3, 3], Twc[:3, 3] + Twc[:3, 2], Twc[:3, 1]
    cells = [(sx, sy, sz) for sz in range(3) for sy in range(3) for sx in range(3)]
    trans = [H.calculate_translation(Twc, grid, *c) for c in cells]
    mvps = np.stack([capi.render_mvp(rp, pos, look, up, t) for t in trans])
    truth_cell = (2, 0, 1)
    t_true = H.calculate_translation(Twc, grid, *truth_cell) * np.float32(0.9)   # near, not on, that cell
    with nmi.NmiContext(w, h) as ctx:
        dx, dr = torch.from_numpy(xyz).cuda(), torch.from_numpy(red).cuda()
        rs = ctx.render_points(dx, dr, mvps, 3.0)
        frame = ctx.render_points(dx, torch.sqrt(dr), capi.render_mvp(rp, pos, look, up, t_true)[None], 3.0)  # other "modality"
        fr = torch.flip(frame[0], dims=[0]).contiguous()       # the camera frame is top-down; renders are bottom-up
        t = torch.zeros(1, 27, device="cuda")
        idx, best = ctx.search_grid(rs, fr[None], t)
    assert cells[idx] == truth_cell and best > 0.3
    tab = t.cpu().numpy().reshape(-1)
    assert best > 1.5 * np.sort(tab)[-2]


@pytest.mark.gpu
def test_gpu_level_graph_equals_separate_calls():
    """nmi_level_* (renders + warps + search + winner as one captured HIP graph) gives exactly what the three calls give,
    replay after replay with changing matrices."""
    torch = pytest.importorskip("torch")
    import orbslam2_nmi_amd as nmi
    from orbslam2_nmi_amd import hostapi as H
    w, h = 160, 120
    xyz, red, rp = plane_cloud(w, h, density=2.0)
    Twc = np.eye(4, dtype=np.float32)
    Twc[:3, 1] = [0, -1, 0]
    pos, look, up = Twc[:3, 3], Twc[:3, 3] + Twc[:3, 2], Twc[:3, 1]
    cells = [(sx, sy, sz) for sz in range(2) for sy in range(2) for sx in range(2)]
    K = sy.intrinsics(w, h)
    with nmi.NmiContext(w, h) as ctx:
        dx, dr = torch.from_numpy(xyz).cuda(), torch.from_numpy(red).cuda()
        frame = torch.flip(ctx.render_points(dx, torch.sqrt(dr), capi.render_mvp(rp, pos, look, up, (0.05, 0, 0))[None], 3.0)[0],
                           dims=[0]).contiguous()
        with nmi.NmiLevel(ctx, dx, dr, frame, 8, 12, 3.0) as lv:
            for lvl in range(4):
                g = H.SearchKernel.make([2, 2, 2, 3, 2, 2], [s / 2 ** lvl for s in (0.2, 0.2, 0.5, 0.02, 0.02, 0.05)])
                mvps = np.stack([capi.render_mvp(rp, pos, look, up, H.calculate_translation(Twc, g, *c)) for c in cells])
                Ms = capi.warp_homographies(K, (3, 2, 2), tuple(g.step[3:6]))
                got = lv.run(mvps, Ms)
                rs = ctx.render_points(dx, dr, mvps, 3.0)
                ws = ctx.warp_stack(frame, Ms)
                assert got == ctx.search_grid(rs, ws), lvl


@pytest.mark.gpu
def test_gpu_level_without_common_planes_lists_every_wavefront():
    """One view matrix that cannot be inverted: the host finds no planes around the views (six zero planes), the prep kernel then
    lists EVERY wavefront of the cloud for the front kernel's workers, and the renders are still nmi_render_points' -- the
    unusable view included (all its points are clipped either way)."""
    torch = pytest.importorskip("torch")
    import orbslam2_nmi_amd as nmi
    w, h, S, Wn = 160, 120, 4, 1
    rng = np.random.default_rng(5)
    K = sy.intrinsics(w, h)
    rp = capi.RenderParams(fx=K[0, 0], fy=K[1, 1], cx=K[0, 2], cy=K[1, 2], near_plane=0.5, far_plane=40.0, point_size=2.0)
    n = 70_001   # (not a multiple of 64: the last wavefront is ragged)
    xyz = (rng.uniform(-1, 1, (n, 3)) * [12, 9, 12] + [0, 0, 8]).astype(np.float32)
    red = rng.uniform(0, 1, n).astype(np.float32)
    mvps = np.stack([capi.render_mvp(rp, (0.1 * s, 0, 0), (0.1 * s, 0.05 * s, 1), (0, -1, 0), (0, 0, 0)) for s in range(S)])
    mvps[2] = 0.0
    Ms = capi.warp_homographies(K, (Wn, 1, 1), (0.02, 0.02, 0.05))
    with nmi.NmiContext(w, h) as ctx:
        dx, dr = torch.from_numpy(xyz).cuda(), torch.from_numpy(red).cuda()
        frame = ctx.render_points(dx, dr, mvps[:1], 2.0)[0].contiguous()
        want = ctx.render_points(dx, dr, mvps, 2.0).cpu().numpy()
        assert (want[2] == 255).all() and (want[0] != 255).any()
        with nmi.NmiLevel(ctx, dx, dr, frame, S, Wn, 2.0) as lv:
            for rep in range(3):
                lv.run(mvps, Ms)
                assert np.array_equal(lv.outputs()[0], want), rep


@pytest.mark.gpu
@pytest.mark.parametrize("seed", [11, 12, 13])
def test_gpu_level_cull_by_common_planes_changes_nothing(seed):
    """A level's front kernel first tests a wavefront's box against six planes around ALL the views' frusta (made by the host per
    replay, level_views_bound).  Views that look in different directions from different places, a cloud that fills a volume far
    larger than any frustum (most wavefronts are culled, many straddle a frustum's side): the level's renders must be exactly
    nmi_render_points' (which has no such test)."""
    torch = pytest.importorskip("torch")
    import orbslam2_nmi_amd as nmi
    w, h, S, Wn = 160, 120, 9, 2
    rng = np.random.default_rng(seed)
    K = sy.intrinsics(w, h)
    rp = capi.RenderParams(fx=K[0, 0], fy=K[1, 1], cx=K[0, 2], cy=K[1, 2], near_plane=0.5, far_plane=40.0, point_size=3.0)
    n = 200_000
    xyz = (rng.uniform(-1, 1, (n, 3)) * [12, 9, 12] + [0, 0, 8]).astype(np.float32)
    xyz = xyz[np.lexsort((xyz[:, 0], xyz[:, 1], xyz[:, 2]))]   # wavefronts = compact runs, as a sorted map has them
    red = rng.uniform(0, 1, n).astype(np.float32)
    mvps = []
    for _ in range(S):
        pos = rng.uniform(-1.5, 1.5, 3)
        look = pos + np.array([rng.uniform(-0.6, 0.6), rng.uniform(-0.4, 0.4), 1.0])
        mvps.append(capi.render_mvp(rp, pos, look, (0, -1, 0), rng.uniform(-0.3, 0.3, 3)))
    mvps = np.stack(mvps)
    Ms = capi.warp_homographies(K, (Wn, 1, 1), (0.02, 0.02, 0.05))
    with nmi.NmiContext(w, h) as ctx:
        dx, dr = torch.from_numpy(xyz).cuda(), torch.from_numpy(red).cuda()
        frame = ctx.render_points(dx, dr, mvps[:1], 3.0)[0].contiguous()
        want = ctx.render_points(dx, dr, mvps, 3.0).cpu().numpy()
        assert 0.2 < float((want != 255).mean()) < 1.0   # the views see the cloud, and not only the cloud
        with nmi.NmiLevel(ctx, dx, dr, frame, S, Wn, 3.0) as lv:
            for rep in range(2):   # (the double-buffered anchors: both parities)
                lv.run(mvps, Ms)
                renders = lv.outputs()[0]
                assert np.array_equal(renders, want), (seed, rep, int((renders != want).sum()))


# ---- textured-mesh mode (nmi_prop_RENDER 1) --------------------------------------------------------------------------
from oracle import mesh_oracle_np as mo  # noqa: E402


def test_mip_chain_and_luma():
    rng = np.random.default_rng(1)
    rgb = rng.integers(0, 256, (8, 16, 3), dtype=np.uint8)
    lv = mo.mip_luma(rgb)
    assert [l.shape for l in lv] == [(8, 16), (4, 8), (2, 4), (1, 2), (1, 1)]
    exp0 = 0.299 * rgb[..., 0] / 255 + 0.587 * rgb[..., 1] / 255 + 0.114 * rgb[..., 2] / 255   # byte 0 weighs 0.299 (shader :16)
    assert np.allclose(lv[0], exp0, atol=1e-6)
    box = (rgb[0:2, 0:2].astype(int).sum(axis=(0, 1)) + 2) // 4
    assert np.allclose(lv[1][0, 0], 0.299 * box[0] / 255 + 0.587 * box[1] / 255 + 0.114 * box[2] / 255, atol=1e-6)


def plane_mesh(w, h, depth=10.0, nx=24, ny=18, seed=3):
    """A textured plane as nx x ny quads (two counter-clockwise triangles each, seen from the origin looking along +z with
    up = -y), plus an occluder nearer to the camera, a back-facing triangle and one partly outside the frustum."""
    rp = params(w, h)
    us = np.linspace(-0.3 * w, 1.3 * w, nx + 1)
    vs = np.linspace(-0.3 * h, 1.3 * h, ny + 1)
    def world(u, v, d):
        return [(u - rp.cx) / rp.fx * d, (v - rp.cy) / rp.fy * d, d]
    tris, uvs = [], []
    for j in range(ny):
        for i in range(nx):
            p00, p10, p01, p11 = world(us[i], vs[j], depth), world(us[i + 1], vs[j], depth), world(us[i], vs[j + 1], depth), world(us[i + 1], vs[j + 1], depth)
            t00, t10, t01, t11 = (i / nx, j / ny), ((i + 1) / nx, j / ny), (i / nx, (j + 1) / ny), ((i + 1) / nx, (j + 1) / ny)
            tris += [p00, p10, p11, p00, p11, p01]
            uvs += [t00, t10, t11, t00, t11, t01]
    tris += [world(0.3 * w, 0.3 * h, 7.0), world(0.6 * w, 0.35 * h, 7.0), world(0.45 * w, 0.7 * h, 8.0)]       # occluder
    uvs += [(0.1, 0.1), (2.4, 0.2), (1.2, 2.9)]                                                                 # repeats (GL_REPEAT), minified
    tris += [world(0.7 * w, 0.2 * h, 6.0), world(0.8 * w, 0.4 * h, 6.0), world(0.9 * w, 0.2 * h, 6.0)]          # opposite winding
    uvs += [(0, 0), (1, 1), (1, 0)]
    tris += [world(0.9 * w, 0.8 * h, 9.0), world(1.6 * w, 0.85 * h, 9.0), world(1.2 * w, 1.5 * h, 9.0)]         # leaves the frustum
    uvs += [(0, 0), (1, 0), (0.5, 1)]
    # the camera of this test mirrors x (rendering.hpp:196-202 puts fx / -cx on the diagonal, and up = -y): corner order as
    # written above is clockwise on screen, so reverse every triangle to make the plane front-facing
    xyz = np.array(tris, np.float32).reshape(-1, 3, 3)[:, ::-1].reshape(-1, 3).copy()
    uv = np.array(uvs, np.float32).reshape(-1, 3, 2)[:, ::-1].reshape(-1, 2).copy()
    B = sy.scene(128, 128, seed)
    rgb = np.stack([B, np.roll(B, 7, 0), np.roll(B, 11, 1)], -1).astype(np.uint8)
    return xyz, uv, rgb, rp


def test_mesh_twin_sanity():
    w, h = 96, 72
    xyz, uv, rgb, rp = plane_mesh(w, h)
    lv = mo.mip_luma(rgb)
    m = capi.render_mvp(rp, (0, 0, 0), (0, 0, 1), (0, -1, 0), (0, 0, 0))
    img = mo.render_mesh(xyz, uv, lv, m, w, h)
    assert (img != 255).mean() > 0.97                       # the plane covers the view (quads face the camera)
    flipped = xyz.reshape(-1, 3, 3)[:, ::-1].reshape(-1, 3)  # reverse every winding: everything is culled but the one back-facer
    img2 = mo.render_mesh(flipped, uv.reshape(-1, 3, 2)[:, ::-1].reshape(-1, 2), lv, m, w, h)
    assert 0 < (img2 != 255).mean() < 0.05


def ground_mesh(w, h, y_ground=4.0, z_back=-30.0, z_front=80.0, half_width=60.0, strips=3):
    """A ground plane BELOW the camera (camera at the origin looking along +z, up = -y, so larger y is lower) that starts far
    behind the camera and runs past the far plane: `strips` x 2 large triangles, every one of them crossing the near plane
    (and the eye plane w = 0).  Plus a wall next to the camera that also reaches behind it."""
    rp = params(w, h)
    xs = np.linspace(-half_width, half_width, strips + 1)
    tris, uvs = [], []
    for i in range(strips):
        a, b = xs[i], xs[i + 1]
        p00, p10, p01, p11 = (a, y_ground, z_back), (b, y_ground, z_back), (a, y_ground, z_front), (b, y_ground, z_front)
        t00, t10, t01, t11 = (i / strips, 0.0), ((i + 1) / strips, 0.0), (i / strips, 4.0), ((i + 1) / strips, 4.0)
        tris += [p00, p10, p11, p00, p11, p01]
        uvs += [t00, t10, t11, t00, t11, t01]
    wall = [(-3.0, -4.0, -10.0), (-3.0, y_ground, -10.0), (-3.0, y_ground, 40.0), (-3.0, -4.0, -10.0), (-3.0, y_ground, 40.0), (-3.0, -4.0, 40.0)]
    tris += wall
    uvs += [(0, 0), (0, 1), (3, 1), (0, 0), (3, 1), (3, 0)]
    xyz = np.array(tris, np.float32)
    uv = np.array(uvs, np.float32)
    B = sy.scene(128, 128, 9)
    rgb = np.stack([B, np.roll(B, 5, 0), np.roll(B, 9, 1)], -1).astype(np.uint8)
    rgb = np.minimum(rgb, 250)  # keep the texture away from the background value
    return xyz, uv, rgb, rp


def _both_windings(xyz, uv):
    """Every triangle twice, once per winding: whichever way the camera's handedness turns out, one copy faces it."""
    x3, u3 = xyz.reshape(-1, 3, 3), uv.reshape(-1, 3, 2)
    return np.concatenate([x3, x3[:, ::-1]]).reshape(-1, 3).copy(), np.concatenate([u3, u3[:, ::-1]]).reshape(-1, 2).copy()


def test_mesh_twin_clips_triangles_at_the_near_plane():
    """GL clips triangles against the view volume (rendering.hpp:294-300,619); a ground plane that passes under the camera
    must cover the image below the horizon instead of vanishing because its corners lie behind the eye."""
    w, h = 96, 72
    xyz, uv, rgb, rp = ground_mesh(w, h)
    xyz, uv = _both_windings(xyz, uv)
    lv = mo.mip_luma(rgb)
    m = capi.render_mvp(rp, (0, 0, 0), (0, 0, 1), (0, -1, 0), (0, 0, 0))
    img = mo.render_mesh(xyz, uv, lv, m, w, h)
    covered = img != 255
    # The ground (4 m from the optical axis) is visible from the far plane (30 m: fy * 4 / 30 pixels off the horizon) to
    # the near plane (5 m: beyond the image edge): one half of the image from a few rows off the horizon to its edge,
    # whichever way this camera's up vector and row order turn it.
    col = covered[:, w // 2]
    off_far = int(np.ceil(rp.fy * 4.0 / rp.far_plane)) + 2
    cy = int(round(rp.cy))
    lower, upper = col[: cy - off_far], col[cy + off_far:]
    assert lower.all() != upper.all() and (lower.all() or upper.all()), col.astype(int)
    ground_rows = slice(0, cy - off_far) if lower.all() else slice(cy + off_far, h)
    sky_rows = slice(cy + off_far, h) if lower.all() else slice(0, cy - off_far)
    assert covered[ground_rows, 8: w - 8].mean() > 0.97       # ... across the width of the image
    edge_row = 0 if lower.all() else h - 1
    assert covered[edge_row].mean() > 0.9                       # including the row next to the near plane's cut
    assert covered[sky_rows].mean() < 0.5                       # the other half: only the wall beside the camera
    # every ground triangle has corners behind the eye: dropped instead of clipped, nothing at all would be drawn
    assert covered.mean() > 0.4


@pytest.mark.gpu
def test_gpu_mesh_clipping_vs_twin():
    """Triangles crossing the near plane / the eye plane: the HIP rasteriser (per-lane and tile-queue kernels) against the
    numpy twin -- identical coverage and depth decisions, at most one grey level on a few pixels (LOD through log2f)."""
    torch = pytest.importorskip("torch")
    import orbslam2_nmi_amd as nmi
    w, h = 320, 200
    gx, gu, rgb, rp = ground_mesh(w, h)
    gx, gu = _both_windings(gx, gu)
    px, pu, _, _ = plane_mesh(w, h, depth=30.0, nx=6, ny=4)
    xyz, uv = np.concatenate([gx, px]), np.concatenate([gu, pu])
    mvps = np.stack([capi.render_mvp(rp, (0, 0, 0), (0, 0, 1), (0, -1, 0), t)
                     for t in ((0, 0, 0), (0.4, -0.3, 1.0), (-0.8, 0.2, -3.0), (2.5, 0.5, 0), (0, -1.0, 2.0))])
    exp = mo.render_stack(xyz, uv, mo.mip_luma(rgb), mvps, w, h)
    assert ((exp != 255).mean(axis=(1, 2)) > 0.4).all()        # the ground is there in every view
    outs = []
    with nmi.NmiContext(w, h) as ctx, nmi.NmiTexture(ctx, rgb) as tex:
        dx, du = torch.from_numpy(xyz).cuda(), torch.from_numpy(uv).cuda()
        for cap, clip_cap in ((4 << 20, 1 << 18), (11, 1 << 18), (0, 1 << 18), (4 << 20, 3), (0, 0)):
            ctx.set_option(ctx.OPT_TILE_QUEUE, cap)              # tile queue / overflowing tile queue / per-lane shading only
            ctx.set_option(ctx.OPT_CLIP_QUEUE, clip_cap)         # clip queue / overflowing: the clip pass rescans the mesh
            outs.append(ctx.render_mesh(dx, du, tex, mvps).cpu().numpy())
    assert all((outs[0] == o).all() for o in outs[1:])
    diff = np.abs(outs[0].astype(int) - exp.astype(int))
    assert ((outs[0] == 255) == (exp == 255)).all()
    assert diff.max() <= 1 and (diff != 0).mean() < 2e-3, (diff.max(), (diff != 0).mean())


@pytest.mark.gpu
def test_gpu_mesh_vs_twin():
    torch = pytest.importorskip("torch")
    import orbslam2_nmi_amd as nmi
    w, h = 160, 120
    xyz, uv, rgb, rp = plane_mesh(w, h)
    mvps = np.stack([capi.render_mvp(rp, (0, 0, 0), (0, 0, 1), (0, -1, 0), t) for t in ((0, 0, 0), (0.4, -0.3, 1.0), (-0.8, 0.2, -3.0))])
    with nmi.NmiContext(w, h) as ctx, nmi.NmiTexture(ctx, rgb) as tex:
        got = ctx.render_mesh(torch.from_numpy(xyz).cuda(), torch.from_numpy(uv).cuda(), tex, mvps).cpu().numpy()
    exp = mo.render_stack(xyz, uv, mo.mip_luma(rgb), mvps, w, h)
    assert got.shape == exp.shape
    diff = np.abs(got.astype(int) - exp.astype(int))
    # coverage / depth decisions identical; the texture LOD goes through log2f, whose last bit may differ between the
    # device library and numpy, so a handful of pixels may differ by one grey level
    assert ((got == 255) == (exp == 255)).all()
    assert diff.max() <= 1 and (diff != 0).mean() < 2e-3, (diff.max(), (diff != 0).mean())


@pytest.mark.gpu
def test_gpu_mesh_large_triangles_and_queue_overflow():
    """Large triangles go through the (triangle, view, tile) queue and the tile kernel; with the queue shortened so that it
    overflows, or switched off, the lanes of the first pass shade the excess themselves.  All three give the same images,
    equal to the twin's: a plane of 6 x 4 quads (triangles of ~1,000 pixels spanning several 64 x 64 tiles), seen from 5 poses."""
    torch = pytest.importorskip("torch")
    import orbslam2_nmi_amd as nmi
    w, h = 320, 200
    xyz, uv, rgb, rp = plane_mesh(w, h, nx=6, ny=4)
    mvps = np.stack([capi.render_mvp(rp, (0, 0, 0), (0, 0, 1), (0, -1, 0), t)
                     for t in ((0, 0, 0), (0.4, -0.3, 1.0), (-0.8, 0.2, -3.0), (2.5, 0, 0), (0, 1.5, 2.0))])
    exp = mo.render_stack(xyz, uv, mo.mip_luma(rgb), mvps, w, h)
    outs = []
    with nmi.NmiContext(w, h) as ctx, nmi.NmiTexture(ctx, rgb) as tex:
        dx, du = torch.from_numpy(xyz).cuda(), torch.from_numpy(uv).cuda()
        for cap in (4 << 20, 37, 0):
            ctx.set_option(ctx.OPT_TILE_QUEUE, cap)
            outs.append(ctx.render_mesh(dx, du, tex, mvps).cpu().numpy())
    assert (outs[0] == outs[1]).all() and (outs[0] == outs[2]).all()
    diff = np.abs(outs[0].astype(int) - exp.astype(int))
    assert ((outs[0] == 255) == (exp == 255)).all()
    assert diff.max() <= 1 and (diff != 0).mean() < 2e-3, (diff.max(), (diff != 0).mean())


def _coplanar_scene(w, h):
    """Two copies of a textured plane at the SAME depth with different uv (so every pixel is an exact depth tie between two
    triangles), the second copy cut into many small triangles: ties between large (binned) and small (per-lane) triangles."""
    xa, ua, rgb, rp = plane_mesh(w, h, nx=3, ny=2)
    xa, ua = xa[:3 * 2 * 3 * 2], ua[:3 * 2 * 3 * 2]                    # the plane only (no occluder / back-facer)
    xb, ub, _, _ = plane_mesh(w, h, nx=60, ny=44)
    xb, ub = xb[:3 * 2 * 60 * 44], ub[:3 * 2 * 60 * 44]
    ub = (ub + np.float32(0.37)).astype(np.float32)                    # another part of the texture
    return xa, ua, xb, ub, rgb, rp


def test_mesh_twin_equal_depth_goes_to_the_triangle_drawn_first():
    w, h = 96, 72
    xa, ua, xb, ub, rgb, rp = _coplanar_scene(w, h)
    lv = mo.mip_luma(rgb)
    m = capi.render_mvp(rp, (0, 0, 0), (0, 0, 1), (0, -1, 0), (0, 0, 0))
    only_a, only_b = mo.render_mesh(xa, ua, lv, m, w, h), mo.render_mesh(xb, ub, lv, m, w, h)
    ab = mo.render_mesh(np.concatenate([xa, xb]), np.concatenate([ua, ub]), lv, m, w, h)
    ba = mo.render_mesh(np.concatenate([xb, xa]), np.concatenate([ub, ua]), lv, m, w, h)
    assert (only_a != only_b).mean() > 0.5
    # glDepthFunc(GL_LESS): a later fragment of equal depth does not replace the earlier one.  Depths of the two copies are
    # interpolated from different corners, so they agree on most, not all, pixels: where they do, draw order decides.
    assert (ab == only_a).mean() > 0.6 and (ba == only_b).mean() > 0.6 and (ab != ba).mean() > 0.3


@pytest.mark.gpu
def test_gpu_mesh_equal_depth_goes_to_the_triangle_drawn_first():
    """The same on the device, bit for bit against the twin, whichever way the fragments travel: bins (large triangles),
    the per-lane path (small ones), overflowing bins -- and run to run (the visibility key orders equal depths by triangle
    index, so the order of the atomics does not show)."""
    torch = pytest.importorskip("torch")
    import orbslam2_nmi_amd as nmi
    w, h = 160, 120
    xa, ua, xb, ub, rgb, rp = _coplanar_scene(w, h)
    mvps = np.stack([capi.render_mvp(rp, (0, 0, 0), (0, 0, 1), (0, -1, 0), t) for t in ((0, 0, 0), (0.3, -0.2, 0.5))])
    lv = mo.mip_luma(rgb)
    with nmi.NmiContext(w, h) as ctx, nmi.NmiTexture(ctx, rgb) as tex:
        for first, second in (((xa, ua), (xb, ub)), ((xb, ub), (xa, ua))):
            xyz, uv = np.concatenate([first[0], second[0]]), np.concatenate([first[1], second[1]])
            exp = mo.render_stack(xyz, uv, lv, mvps, w, h)
            dx, du = torch.from_numpy(xyz).cuda(), torch.from_numpy(uv).cuda()
            outs = []
            for cap in (255, 255, 3, 0):
                ctx.set_option(ctx.OPT_TILE_QUEUE, cap)
                outs.append(ctx.render_mesh(dx, du, tex, mvps).cpu().numpy())
            assert all((outs[0] == o).all() for o in outs[1:])
            diff = np.abs(outs[0].astype(int) - exp.astype(int))
            assert ((outs[0] == 255) == (exp == 255)).all()
            assert diff.max() <= 1 and (diff != 0).mean() < 2e-3, (diff.max(), (diff != 0).mean())


@pytest.mark.gpu
@pytest.mark.parametrize("seed", [1, 2, 3])
def test_gpu_mesh_random_soup_vs_twin(seed):
    """A soup of random triangles -- pixel-sized to screen-filling, every orientation, some crossing the near plane or the
    frustum's sides, many overlapping at different depths -- through bins, the per-lane path and overflowing bins: the three
    give one image, and it is the twin's (coverage, depth and draw-order decisions identical; <= 1 grey level through the LOD)."""
    torch = pytest.importorskip("torch")
    import orbslam2_nmi_amd as nmi
    w, h = 160, 120
    rng = np.random.default_rng(seed)
    rp = params(w, h)
    n = 140
    depth = rng.uniform(3.0, 28.0, (n, 1))                       # near plane 5, far plane 30: some in front of the near plane
    centre_px = rng.uniform([-0.4 * w, -0.4 * h], [1.4 * w, 1.4 * h], (n, 2))
    size_px = np.exp(rng.uniform(np.log(0.6), np.log(0.5 * w), (n, 1)))
    corners = centre_px[:, None, :] + size_px[:, None, :] * rng.uniform(-1, 1, (n, 3, 2))
    z = depth[:, None, :] * rng.uniform(0.7, 1.4, (n, 3, 1))     # tilted: depth varies across a triangle
    xyz = np.concatenate([(corners[..., :1] - rp.cx) / rp.fx * z, (corners[..., 1:] - rp.cy) / rp.fy * z, z], -1).astype(np.float32)
    uv = rng.uniform(-1.5, 2.5, (n, 3, 2)).astype(np.float32)
    xyz, uv = _both_windings(xyz.reshape(-1, 3), uv.reshape(-1, 2))
    B = sy.scene(128, 64, 17 + seed)
    rgb = np.stack([B, np.roll(B, 3, 0), np.roll(B, 7, 1)], -1).astype(np.uint8)
    mvps = np.stack([capi.render_mvp(rp, (0, 0, 0), (0, 0, 1), (0, -1, 0), t) for t in ((0, 0, 0), (0.6, -0.4, 1.5), (-1.0, 0.3, -2.0))])
    exp = mo.render_stack(xyz, uv, mo.mip_luma(rgb), mvps, w, h)
    assert 0.15 < (exp != 255).mean() < 0.999         # covered and background both present
    outs = []
    with nmi.NmiContext(w, h) as ctx, nmi.NmiTexture(ctx, rgb) as tex:
        dx, du = torch.from_numpy(xyz).cuda(), torch.from_numpy(uv).cuda()
        for cap, clip_cap in ((255, 1 << 18), (255, 1 << 18), (5, 1 << 18), (0, 1 << 18), (255, 2)):
            ctx.set_option(ctx.OPT_TILE_QUEUE, cap)
            ctx.set_option(ctx.OPT_CLIP_QUEUE, clip_cap)
            outs.append(ctx.render_mesh(dx, du, tex, mvps).cpu().numpy())
    assert all((outs[0] == o).all() for o in outs[1:])
    diff = np.abs(outs[0].astype(int) - exp.astype(int))
    assert ((outs[0] == 255) == (exp == 255)).all()
    assert diff.max() <= 1 and (diff != 0).mean() < 2e-3, (diff.max(), (diff != 0).mean())


@pytest.mark.gpu
def test_gpu_mesh_to_winner_end_to_end():
    """mesh + texture + pose -> device render stack for a 3x3x1 translation grid; the frame is the mesh seen from a
    displaced pose (other gamma + noise); the search picks the nearest cell."""
    torch = pytest.importorskip("torch")
    import orbslam2_nmi_amd as nmi
    from orbslam2_nmi_amd import hostapi as H
    w, h = 320, 240
    xyz, uv, rgb, rp = plane_mesh(w, h, nx=40, ny=30)
    Twc = np.eye(4, dtype=np.float32)
    Twc[:3, 1] = [0, -1, 0]
    grid = H.SearchKernel.make([3, 3, 1, 1, 1, 1], [0.3, 0.3, 0.5, 0.02, 0.02, 0.05])
    pos, look, up = Twc[:3, 3], Twc[:3, 3] + Twc[:3, 2], Twc[:3, 1]
    cells = [(sx, sy_, 0) for sy_ in range(3) for sx in range(3)]
    mvps = np.stack([capi.render_mvp(rp, pos, look, up, H.calculate_translation(Twc, grid, *c)) for c in cells])
    truth = (0, 2, 0)
    t_true = H.calculate_translation(Twc, grid, *truth) * np.float32(0.85)
    with nmi.NmiContext(w, h) as ctx, nmi.NmiTexture(ctx, rgb) as tex:
        dx, du = torch.from_numpy(xyz).cuda(), torch.from_numpy(uv).cuda()
        rs = ctx.render_mesh(dx, du, tex, mvps)
        fr = ctx.render_mesh(dx, du, tex, capi.render_mvp(rp, pos, look, up, t_true)[None])[0]
        noise = torch.from_numpy(np.random.default_rng(2).normal(0, 6, (h, w)).astype(np.float32)).cuda()
        frame = torch.clamp(torch.round(255.0 * (torch.flip(fr, dims=[0]).float() / 255.0) ** 0.7 + noise), 0, 255).to(torch.uint8).contiguous()
        t = torch.zeros(1, 9, device="cuda")
        idx, best = ctx.search_grid(rs, frame[None], t)
    assert cells[idx] == truth and best > 1.3 * np.sort(t.cpu().numpy().reshape(-1))[-2]


@pytest.mark.gpu
def test_gpu_mesh_level_graph_equals_separate_calls():
    """nmi_level_create_mesh: mesh renders + warps + search + winner as one captured HIP graph give exactly what the three
    calls give, replay after replay with changing matrices (triangles large enough to take the tile queue)."""
    torch = pytest.importorskip("torch")
    import orbslam2_nmi_amd as nmi
    from orbslam2_nmi_amd import hostapi as H
    w, h = 160, 120
    xyz, uv, rgb, rp = plane_mesh(w, h, nx=12, ny=9)
    Twc = np.eye(4, dtype=np.float32)
    Twc[:3, 1] = [0, -1, 0]
    pos, look, up = Twc[:3, 3], Twc[:3, 3] + Twc[:3, 2], Twc[:3, 1]
    cells = [(sx, sy_, sz) for sz in range(2) for sy_ in range(2) for sx in range(2)]
    K = sy.intrinsics(w, h)
    with nmi.NmiContext(w, h) as ctx, nmi.NmiTexture(ctx, rgb) as tex:
        dx, du = torch.from_numpy(xyz).cuda(), torch.from_numpy(uv).cuda()
        fr = ctx.render_mesh(dx, du, tex, capi.render_mvp(rp, pos, look, up, (0.05, 0, 0))[None])[0]
        frame = torch.flip(fr, dims=[0]).contiguous()
        with nmi.NmiLevel(ctx, dx, du, frame, 8, 12, 1.0, texture=tex) as lv:
            for lvl in range(4):
                g = H.SearchKernel.make([2, 2, 2, 3, 2, 2], [s / 2 ** lvl for s in (0.2, 0.2, 0.5, 0.02, 0.02, 0.05)])
                mvps = np.stack([capi.render_mvp(rp, pos, look, up, H.calculate_translation(Twc, g, *c)) for c in cells])
                Ms = capi.warp_homographies(K, (3, 2, 2), tuple(g.step[3:6]))
                got = lv.run(mvps, Ms)
                rs = ctx.render_mesh(dx, du, tex, mvps)
                ws = ctx.warp_stack(frame, Ms)
                assert got == ctx.search_grid(rs, ws), lvl


@pytest.mark.gpu
def test_gpu_sorted_maps_are_permutations_and_render_identically():
    """nmi_sort_points / nmi_sort_triangles (Morton order, device radix sort): the output is a permutation of the input
    records (NaN / inf positions last), neighbours in memory are neighbours in space, and the renders do not change."""
    torch = pytest.importorskip("torch")
    import orbslam2_nmi_amd as nmi
    w, h = 160, 120
    rng = np.random.default_rng(5)
    views = np.stack([capi.render_mvp(params(w, h), (0.1 * k, 0, 0), (0.1 * k, 0, 1), (0, -1, 0), (0, 0, 0.2 * k)) for k in range(3)])  # noqa
    with nmi.NmiContext(w, h) as ctx:
        # points
        xyz, red, rp = plane_cloud(w, h, density=3.0)
        perm = rng.permutation(len(xyz))
        xyz, red = xyz[perm].copy(), red[perm].copy()
        xyz[5] = [np.nan, 0, 10]
        xyz[77] = [0, np.inf, 10]
        dx, dr = torch.from_numpy(xyz).cuda(), torch.from_numpy(red).cuda()
        sx, sr = ctx.sort_points(dx, dr)
        a = np.concatenate([xyz, red[:, None]], 1)
        b = np.concatenate([sx.cpu().numpy(), sr.cpu().numpy()[:, None]], 1)
        key = lambda m: m[np.lexsort(np.nan_to_num(m, nan=1e30, posinf=2e30, neginf=-2e30).T[::-1])]
        assert np.array_equal(key(a), key(b), equal_nan=True)                   # same multiset of records
        assert not np.isfinite(b[-2:, :3]).all(1).any() and np.isfinite(b[:-2, :3]).all()  # the two bad points went last
        step_in = np.linalg.norm(np.diff(xyz[np.isfinite(xyz).all(1)], axis=0), axis=1).mean()
        step_out = np.linalg.norm(np.diff(b[:-2, :3], axis=0), axis=1).mean()
        assert step_out < 0.1 * step_in                                          # consecutive points are now close
        r0 = ctx.render_points(dx, dr, views, 3.0).cpu().numpy()
        r1 = ctx.render_points(sx, sr, views, 3.0).cpu().numpy()
        assert (r0 == r1).all() and (r0 != 255).mean() > 0.3
        # triangles
        txyz, tuv, tex, _ = plane_mesh(w, h)
        t = len(txyz) // 3
        tp = rng.permutation(t)
        txyz = txyz.reshape(t, 9)[tp].reshape(-1, 3).copy()
        tuv = tuv.reshape(t, 6)[tp].reshape(-1, 2).copy()
        dtx, dtu = torch.from_numpy(txyz).cuda(), torch.from_numpy(tuv).cuda()
        stx, stu = ctx.sort_triangles(dtx, dtu)
        a = np.concatenate([txyz.reshape(t, 9), tuv.reshape(t, 6)], 1)
        b = np.concatenate([stx.cpu().numpy().reshape(t, 9), stu.cpu().numpy().reshape(t, 6)], 1)
        assert np.array_equal(key(a), key(b))
        with nmi.NmiTexture(ctx, tex) as texture:
            m0 = ctx.render_mesh(dtx, dtu, texture, views).cpu().numpy()
            m1 = ctx.render_mesh(stx, stu, texture, views).cpu().numpy()
        assert (m0 == m1).all() and (m0 != 255).mean() > 0.3
        # empty maps and aliasing are handled
        e = torch.empty((0, 3), dtype=torch.float32, device="cuda")
        ex, er = ctx.sort_points(e, torch.empty(0, dtype=torch.float32, device="cuda"))
        assert ex.shape[0] == 0


# ------------------------------------------------------------------------------------------------------------------
# The short reciprocal of the producers (nmi_warp_device.h: hardware approximation + one Newton step instead of the
# compiler's 11-instruction division) must have the division's bits: checked for every float on the GPU at hand.

@pytest.mark.gpu
def test_gpu_reciprocal_exhaustive():
    import os
    import subprocess
    from conftest import ROOT
    exe = os.path.join(ROOT, "examples", "rcp_check")
    if not os.access(exe, os.X_OK):  # normally prebuilt by __graft_entry__.build(); hipcc exists on the GPU box too
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "examples"), "rcp_check"], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    print(r.stdout, r.stderr)
    assert r.returncode == 0 and "RCP OK" in r.stdout, r.stdout + r.stderr


# ------------------------------------------------------------------------------------------------------------------
# The renderers choose between internal forms by the size of the job (binning in one kernel or two, tile kernel with 255 or 127
# entries per tile; a level's cloud through the prep kernel's list).  The choice must not show in a single byte.

@pytest.mark.gpu
def test_gpu_mesh_pass_variants_agree():
    import os
    import subprocess
    import sys
    from conftest import ROOT
    script = os.path.join(ROOT, "tests", "helpers", "mesh_variants.py")
    outs = []
    for extra in ({}, {"NMI_MESH_NO_PAIRS": "1", "NMI_MESH_NO_SMALL_TILES": "1"}, {"NMI_MESH_NO_PAIRS": "1"}, {"NMI_MESH_NO_SMALL_TILES": "1"}):
        r = subprocess.run([sys.executable, script], capture_output=True, text=True, timeout=600, env=dict(os.environ, **extra))
        assert r.returncode == 0, r.stdout + r.stderr
        lines = [l for l in r.stdout.splitlines() if l.startswith("MESH")]
        assert len(lines) == 5, r.stdout + r.stderr
        outs.append(lines)
    assert all(o == outs[0] for o in outs[1:]), outs


@pytest.mark.gpu
def test_gpu_level_cloud_odd_sizes_equal_render_points():
    """Clouds of 1 .. 1,000 points (fewer points than a wavefront, ragged last wavefront, fewer wavefronts than splat workers), 1 and
    64 views, sprite sizes 1, 2 and 5: a level's renders are nmi_render_points'."""
    torch = pytest.importorskip("torch")
    import orbslam2_nmi_amd as nmi
    w, h = 160, 120
    K = sy.intrinsics(w, h)
    rng = np.random.default_rng(1)
    for n in (1, 63, 65, 1000):
        for S in (1, 64):
            for size in (1.0, 2.0, 5.0):
                rp = capi.RenderParams(fx=K[0, 0], fy=K[1, 1], cx=K[0, 2], cy=K[1, 2], near_plane=0.5, far_plane=40.0, point_size=size)
                xyz = (rng.uniform(-1, 1, (n, 3)) * [6, 4, 6] + [0, 0, 9]).astype(np.float32)
                red = rng.uniform(0, 1, n).astype(np.float32)
                mvps = np.stack([capi.render_mvp(rp, rng.uniform(-1, 1, 3), rng.uniform(-1, 1, 3) * [0.3, 0.3, 0] + [0, 0, 4], (0, -1, 0), (0, 0, 0))
                                 for _ in range(S)])
                Ms = capi.warp_homographies(K, (1, 1, 1), (0.02, 0.02, 0.05))
                with nmi.NmiContext(w, h) as ctx:
                    dx, dr = torch.from_numpy(xyz).cuda(), torch.from_numpy(red).cuda()
                    frame = ctx.render_points(dx, dr, mvps[:1], size)[0].contiguous()
                    want = ctx.render_points(dx, dr, mvps, size).cpu().numpy()
                    with nmi.NmiLevel(ctx, dx, dr, frame, S, 1, size) as lv:
                        for rep in range(2):
                            lv.run(mvps, Ms)
                            assert np.array_equal(lv.outputs()[0], want), (n, S, size, rep)
