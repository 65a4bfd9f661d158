"""BASELINE.json configs[4] (streaming: keyframes x 3 coarse-to-fine levels x 3^6 candidates, 848x480) at full frame size.

The loop being streamed is src/Tracking.cc:2088-2130 (<= 4 levels around RelocalizeWithNMI, :1851-1985).  Every level of
every keyframe is checked against the C oracle on the SAME stacks -- winner (index and score) and the whole rating table,
with == (oracle in its rounded term mode):
  * nmi_stream_*: render stacks are host inputs; the warp stack is restated on the host by the fp32 numpy twin of the
    warp producer (oracle/warp_oracle_np.py), so a wrong device warp would show up as a different table;
  * nmi_level_* (one captured HIP graph per level: renders + warps + search on the device): the stacks the graph produced
    are pulled back from the device and handed to the oracle.
GPU tier; the multi-rank orchestration of the same loop is covered on the CPU by tests/test_sharding_gloo.py."""
import numpy as np
import pytest

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu

W, H, COUNTS, LEVELS, KEYFRAMES = 848, 480, (3, 3, 3), 3, 2


def _levels():
    from orbslam2_nmi_amd import capi, synthetic as sy
    K = sy.intrinsics(W, H)
    out = []
    for kf in range(KEYFRAMES):
        B = sy.scene(W, H, 9000 + kf)
        F = sy.camera_frame(B, 9500 + kf)
        for lvl in range(LEVELS):  # steps halved per level like NmiSearchKernel::resizeKernel (nmiSearchKernel.cpp:104-117)
            rs = sy.render_stack(B, COUNTS, shift_px=max(1, 4 >> lvl), zoom_step=0.02 / 2 ** lvl)
            Ms = capi.warp_homographies(K, COUNTS, tuple(s / 2 ** lvl for s in (0.02, 0.02, 0.05)))
            out.append((kf, lvl, F, rs, Ms))
    return out


def test_stream_pipeline_at_size_every_level_equals_the_oracle():
    import orbslam2_nmi_amd as nmi
    from oracle import binding as oc
    from oracle import warp_oracle_np as wo
    levels = _levels()
    got = []
    with nmi.NmiContext(W, H, render_bottom_up=False) as ctx:
        with nmi.NmiStream(ctx, 27, 27, depth=2) as st:
            st.keep_ratings(True)
            pending = []

            def collect():
                t = pending.pop(0)
                win = st.wait(t)
                got.append((win, st.ratings(t, 27, 27)))

            for kf, lvl, F, rs, Ms in levels:
                pending.append(st.submit(torch.from_numpy(rs).pin_memory(), torch.from_numpy(F).pin_memory(), Ms))
                if len(pending) == 2:
                    collect()
            while pending:
                collect()
    assert len(got) == KEYFRAMES * LEVELS
    for (kf, lvl, F, rs, Ms), (win, table) in zip(levels, got):
        ws = wo.warp_stack(F, Ms)  # host restatement of the device warp producer
        with oc.rounded():
            ro, io, bo = oc.search_grid(rs, ws, render_bottom_up=False, threads=16)
        assert (table.view(np.uint32) == ro.view(np.uint32)).all(), (kf, lvl, np.abs(table - ro).max())
        assert win == (io, bo), (kf, lvl, win, io, bo)
        assert io == 13 * 27 + 13, (kf, lvl, io)  # the frame was taken at the grid centre


def test_level_graph_at_size_every_level_equals_the_oracle():
    """nmi_level_*: per level 27 renders of a point cloud + 27 warps + the 729-candidate search replay as one HIP graph."""
    import orbslam2_nmi_amd as nmi
    from oracle import binding as oc
    from orbslam2_nmi_amd import capi, hostapi as Hh, synthetic as sy
    K = sy.intrinsics(W, H)
    rp = capi.RenderParams(fx=K[0, 0], fy=K[1, 1], cx=K[0, 2], cy=K[1, 2], near_plane=5.0, far_plane=30.0, point_size=3.0)
    Twc = np.eye(4, dtype=np.float32)
    pos, look, up = Twc[:3, 3], Twc[:3, 3] + Twc[:3, 2], Twc[:3, 1]
    cells = [(sx, sy_, sz) for sz in range(3) for sy_ in range(3) for sx in range(3)]
    grids = [Hh.SearchKernel.make([3] * 6, [s / 2 ** l for s in (0.2, 0.2, 0.5, 0.02, 0.02, 0.05)]) for l in range(LEVELS)]
    for kf in range(KEYFRAMES):
        B = sy.scene(2 * W, 2 * H, 77 + kf)
        nu, nv = int(2.2 * W), int(2.2 * H)  # ~1 M points: a textured plane at 10 m
        uu, vv = np.meshgrid(np.linspace(-W, 2 * W, nu), np.linspace(-H, 2 * H, nv))
        xyz = np.stack([(uu - rp.cx) / rp.fx * 10.0, (vv - rp.cy) / rp.fy * 10.0, np.full_like(uu, 10.0)], -1).reshape(-1, 3).astype(np.float32)
        red = (B[np.clip(((vv + H) / 3 * 2).astype(int), 0, 2 * H - 1), np.clip(((uu + W) / 3 * 2).astype(int), 0, 2 * W - 1)]
               .astype(np.float32) / np.float32(256)).reshape(-1)
        dx, dr = torch.from_numpy(xyz).cuda(), torch.from_numpy(red).cuda()
        with nmi.NmiContext(W, H) as ctx:
            frame = torch.flip(ctx.render_points(dx, torch.sqrt(dr), capi.render_mvp(rp, pos, look, up, (0, 0, 0))[None], 3.0)[0], dims=[0])
            noise = torch.from_numpy(np.random.default_rng(4242 + kf).normal(0.0, 10.0, (H, W)).astype(np.float32)).cuda()
            frame = torch.clamp(torch.round(frame.float() + noise), 0, 255).to(torch.uint8).contiguous()
            torch.cuda.synchronize()
            with nmi.NmiLevel(ctx, dx, dr, frame, 27, 27, 3.0) as level:
                for lvl, g in enumerate(grids):
                    mvps = np.stack([capi.render_mvp(rp, pos, look, up, Hh.calculate_translation(Twc, g, *c)) for c in cells])
                    homs = capi.warp_homographies(K, (3, 3, 3), tuple(g.step[3:6]))
                    win = level.run(mvps, homs)
                    rs, ws, table = level.outputs()  # the stacks this replay produced, pulled back from the device
                    assert rs.min() < 250 and (rs == 255).mean() < 0.5 and ws.max() > 0  # real renders / warps, not blanks
                    with oc.rounded():
                        ro, io, bo = oc.search_grid(rs, ws, render_bottom_up=True, threads=16)
                    assert (table.view(np.uint32) == ro.view(np.uint32)).all(), (kf, lvl, np.abs(table - ro).max())
                    assert win == (io, bo), (kf, lvl, win, io, bo)
                    if lvl == 0:
                        assert io == 13 * 27 + 13  # coarse level: the centre cell wins outright
