import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import orbslam2_nmi_amd as nmi
from orbslam2_nmi_amd import capi, hostapi as H, synthetic as sy
w, h = 848, 480
K = sy.intrinsics(w, h)
rp = capi.RenderParams(fx=K[0, 0], fy=K[1, 1], cx=K[0, 2], cy=K[1, 2], near_plane=5.0, far_plane=30.0, point_size=3.0)
B = sy.scene(2 * w, 2 * h, 77)
nu, nv = int(3 * w * 0.9), int(3 * h * 0.9)
uu, vv = np.meshgrid(np.linspace(-w, 2 * w, nu), np.linspace(-h, 2 * h, nv))
xyz = np.stack([(uu - rp.cx) / rp.fx * 10.0, (vv - rp.cy) / rp.fy * 10.0, np.full_like(uu, 10.0)], -1).reshape(-1, 3).astype(np.float32)
red = (B[np.clip(((vv + h) / 3 * 2).astype(int), 0, 2 * h - 1), np.clip(((uu + w) / 3 * 2).astype(int), 0, 2 * w - 1)].astype(np.float32) / np.float32(256)).reshape(-1)
st = torch.cuda.Stream(); torch.cuda.set_stream(st)
dx, dr = torch.from_numpy(xyz).cuda(), torch.from_numpy(red).cuda()
ctx = nmi.NmiContext(w, h); ctx.set_stream(st.cuda_stream)
Twc = np.eye(4, dtype=np.float32); Twc[:3, 1] = [0, -1, 0]
pos, look, up = Twc[:3, 3], Twc[:3, 3] + Twc[:3, 2], Twc[:3, 1]
g = H.SearchKernel.make([3] * 6, [0.2, 0.2, 0.5, 0.02, 0.02, 0.05])
cells = [(sx, sy_, sz) for sz in range(3) for sy_ in range(3) for sx in range(3)]
mvps = np.stack([capi.render_mvp(rp, pos, look, up, H.calculate_translation(Twc, g, *c)) for c in cells])
homs = capi.warp_homographies(K, (3, 3, 3), tuple(g.step[3:6]))
frame = torch.flip(ctx.render_points(dx, torch.sqrt(dr), capi.render_mvp(rp, pos, look, up, (0, 0, 0))[None], 3.0)[0], dims=[0]).contiguous()
lv = nmi.NmiLevel(ctx, dx, dr, frame, 27, 27, 3.0)
for _ in range(10): lv.run(mvps, homs)
t0 = time.perf_counter()
for _ in range(200): r = lv.run(mvps, homs)
print("level.run", (time.perf_counter() - t0) / 200 * 1e6, "us", r)
rs = torch.empty((27, h, w), dtype=torch.uint8, device="cuda"); ws = torch.empty((27, h, w), dtype=torch.uint8, device="cuda")
def sep():
    ctx.render_points(dx, dr, mvps, 3.0, out=rs, sync=False); ctx.warp_stack(frame, homs, out=ws, sync=False); return ctx.search_grid(rs, ws)
for _ in range(10): sep()
t0 = time.perf_counter()
for _ in range(200): r = sep()
print("separate ", (time.perf_counter() - t0) / 200 * 1e6, "us", r)
