"""Point-cloud renderer against the order of the cloud in memory: scan order, shuffled, shuffled then nmi_sort_points."""
import os, sys, numpy as np, torch
sys.path.insert(0, '.')
import orbslam2_nmi_amd as nmi
from orbslam2_nmi_amd import capi, hostapi as H, synthetic as sy
w, h = 848, 480
K = sy.intrinsics(w, h)
rp = capi.RenderParams(fx=K[0, 0], fy=K[1, 1], cx=K[0, 2], cy=K[1, 2], near_plane=5.0, far_plane=30.0, point_size=3.0)
nu, nv = int(3 * w * 0.9), int(3 * h * 0.9)
uu, vv = np.meshgrid(np.linspace(-w, 2 * w, nu), np.linspace(-h, 2 * h, nv))
xyz = np.stack([(uu - rp.cx) / rp.fx * 10.0, (vv - rp.cy) / rp.fy * 10.0, np.full_like(uu, 10.0)], -1).reshape(-1, 3).astype(np.float32)
red = (np.arange(xyz.shape[0]) % 251 / 256).astype(np.float32)
st = torch.cuda.Stream(); torch.cuda.set_stream(st)
ctx = nmi.NmiContext(w, h); ctx.set_stream(st.cuda_stream)
Twc = np.eye(4, dtype=np.float32)
pos, look, up = Twc[:3, 3], Twc[:3, 3] + Twc[:3, 2], Twc[:3, 1]
g = H.SearchKernel.make([3] * 6, [0.2, 0.2, 0.5, 0.02, 0.02, 0.05])
cells = [(sx, sy_, sz) for sz in range(3) for sy_ in range(3) for sx in range(3)]
mvps = np.stack([capi.render_mvp(rp, pos, look, up, H.calculate_translation(Twc, g, *c)) for c in cells])
def timed(fn, n=30):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(st)
    for _ in range(n): fn()
    e1.record(st); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
outs = []
import time
shuf = np.random.default_rng(1).permutation(xyz.shape[0])
cases = {"scan order": (xyz, red), "shuffled": (xyz[shuf], red[shuf])}
outs = []
for name, (x, r) in list(cases.items()) + [("shuffled, then nmi_sort_points", (None, None))]:
    if x is None:
        sx, sr = torch.from_numpy(np.ascontiguousarray(cases["shuffled"][0])).cuda(), torch.from_numpy(np.ascontiguousarray(cases["shuffled"][1])).cuda()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        dx, dr = ctx.sort_points(sx, sr)
        sort_ms = (time.perf_counter() - t0) * 1e3
        dx, dr = ctx.sort_points(sx, sr)
        t0 = time.perf_counter()
        dx, dr = ctx.sort_points(sx, sr)
        print(f"nmi_sort_points of {xyz.shape[0]} points: {(time.perf_counter() - t0) * 1e3:.2f} ms (first call {sort_ms:.1f} ms)")
    else:
        dx, dr = torch.from_numpy(np.ascontiguousarray(x)).cuda(), torch.from_numpy(np.ascontiguousarray(r)).cuda()
    rs = torch.empty((27, h, w), dtype=torch.uint8, device="cuda")
    us = timed(lambda: ctx.render_points(dx, dr, mvps, 3.0, out=rs, sync=False))
    outs.append(rs.cpu().numpy())
    print(f"{name}: {us:.1f} us (clear + splat + resolve, 27 views {w}x{h})")
print("identical renders:", (outs[0] == outs[1]).all() and (outs[0] == outs[2]).all())
