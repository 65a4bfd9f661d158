#!/usr/bin/env python3
"""Time of nmi_render_mesh (27 views, 848x480) for one textured plane three views wide, at several tessellations: from
façade-like triangles of ~1,000 pixels to sub-pixel ones.  DESIGN.md sections 4 and 6 quote these numbers."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import orbslam2_nmi_amd as nmi
from orbslam2_nmi_amd import capi, synthetic as sy

w, h = 848, 480
K = sy.intrinsics(w, h)
rp = capi.RenderParams(fx=K[0, 0], fy=K[1, 1], cx=K[0, 2], cy=K[1, 2], near_plane=5.0, far_plane=30.0, point_size=1.0)
B = sy.scene(1024, 1024, 5)
rgb = np.stack([B, B, B], -1).astype(np.uint8)
cells = [(sx, sy_, sz) for sz in range(3) for sy_ in range(3) for sx in range(3)]
mvps = np.stack([capi.render_mvp(rp, (0, 0, 0), (0, 0, 1), (0, 1, 0), ((c[0] - 1) * 0.2, (c[1] - 1) * 0.2, (c[2] - 1) * 0.5)) for c in cells])
ctx = nmi.NmiContext(w, h)
tex = nmi.NmiTexture(ctx, rgb)
for nx, ny in ((60, 40), (300, 200), (1200, 800), (2400, 1600)):
    us, vs = np.linspace(-w, 2 * w, nx + 1), np.linspace(-h, 2 * h, ny + 1)
    uu, vv = np.meshgrid(us, vs)
    P = np.stack([(uu - rp.cx) / rp.fx * 10.0, (vv - rp.cy) / rp.fy * 10.0, np.full_like(uu, 10.0)], -1).astype(np.float32)
    T = np.stack([(uu + w) / (3 * w), (vv + h) / (3 * h)], -1).astype(np.float32)
    p00, p10, p01, p11 = P[:-1, :-1], P[:-1, 1:], P[1:, :-1], P[1:, 1:]
    t00, t10, t01, t11 = T[:-1, :-1], T[:-1, 1:], T[1:, :-1], T[1:, 1:]
    for order in (0, 1):  # one of the two windings faces the camera
        if order == 0:
            xyz, uv = np.stack([p00, p10, p11, p00, p11, p01], 2).reshape(-1, 3), np.stack([t00, t10, t11, t00, t11, t01], 2).reshape(-1, 2)
        else:
            xyz, uv = np.stack([p00, p11, p10, p00, p01, p11], 2).reshape(-1, 3), np.stack([t00, t11, t10, t00, t01, t11], 2).reshape(-1, 2)
        dx, du = torch.from_numpy(np.ascontiguousarray(xyz)).cuda(), torch.from_numpy(np.ascontiguousarray(uv)).cuda()
        out = ctx.render_mesh(dx, du, tex, mvps)
        cov = float((out != 255).float().mean())
        if cov < 0.5:
            continue
        for queue in (4 << 20, 0):
            ctx.set_option(ctx.OPT_TILE_QUEUE, queue)
            if queue == 0 and xyz.shape[0] // 3 < 100000:
                n = 1   # the lane-per-triangle form alone takes tens of milliseconds here
            else:
                n = 10
            ctx.render_mesh(dx, du, tex, mvps, out=out)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(n):
                ctx.render_mesh(dx, du, tex, mvps, out=out, sync=False)
            ctx.synchronize()
            dt = (time.perf_counter() - t0) / n
            print(f"{xyz.shape[0] // 3:9d} triangles, 27 views {w}x{h}, tile queue {'on ' if queue else 'off'}: {dt * 1e6:9.1f} us per stack "
                  f"(coverage {cov:.2f})", flush=True)
        ctx.set_option(ctx.OPT_TILE_QUEUE, 4 << 20)
