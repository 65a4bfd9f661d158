"""Renders five meshes (both windings, relief, 5 / 27 / 64 views) and prints a hash of every stack.  Run by
tests/test_render.py::test_gpu_mesh_pass_variants_agree once per choice of the renderer's internal forms (environment switches that
the library reads once per process): the hashes must not depend on the choice."""
import sys, os, hashlib
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
import orbslam2_nmi_amd as nmi
from orbslam2_nmi_amd import capi, synthetic as sy
w, h = 848, 480
K = sy.intrinsics(w, h)
rp = capi.RenderParams(fx=K[0, 0], fy=K[1, 1], cx=K[0, 2], cy=K[1, 2], near_plane=5.0, far_plane=30.0, point_size=1.0)
B = sy.scene(256, 256, 5)
rgb = np.stack([B, B, B], -1).astype(np.uint8)
out = []
for nx, ny, S in ((40, 30, 27), (90, 60, 27), (90, 60, 64), (150, 100, 5), (400, 300, 27)):
    rng = np.random.default_rng(nx)
    mvps = np.stack([capi.render_mvp(rp, (0, 0, 0), (0, 0, 1), (0, 1, 0), tuple(rng.uniform(-0.4, 0.4, 3))) for _ in range(S)])
    us, vs = np.linspace(-w, 2 * w, nx + 1), np.linspace(-h, 2 * h, ny + 1)
    uu, vv = np.meshgrid(us, vs)
    z = 10.0 + 2.0 * np.sin(uu * 0.01) * np.cos(vv * 0.013)
    P = np.stack([(uu - rp.cx) / rp.fx * z, (vv - rp.cy) / rp.fy * z, z], -1).astype(np.float32)
    T = np.stack([(uu + w) / (3 * w), (vv + h) / (3 * h)], -1).astype(np.float32)
    p00, p10, p01, p11 = P[:-1, :-1], P[:-1, 1:], P[1:, :-1], P[1:, 1:]
    t00, t10, t01, t11 = T[:-1, :-1], T[:-1, 1:], T[1:, :-1], T[1:, 1:]
    xyz = np.concatenate([np.stack([p00, p10, p11, p00, p11, p01], 2).reshape(-1, 3), np.stack([p00, p11, p10, p00, p01, p11], 2).reshape(-1, 3)])
    uv = np.concatenate([np.stack([t00, t10, t11, t00, t11, t01], 2).reshape(-1, 2), np.stack([t00, t11, t10, t00, t01, t11], 2).reshape(-1, 2)])
    with nmi.NmiContext(w, h) as ctx, nmi.NmiTexture(ctx, rgb) as tex:
        img = ctx.render_mesh(torch.from_numpy(np.ascontiguousarray(xyz)).cuda(), torch.from_numpy(np.ascontiguousarray(uv)).cuda(), tex, mvps).cpu().numpy()
    out.append((nx, ny, S, xyz.shape[0] // 3, float((img != 255).mean()).__round__(3), hashlib.sha256(img.tobytes()).hexdigest()[:16]))
for o in out: print("MESH", *o)
