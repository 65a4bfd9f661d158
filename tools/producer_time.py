#!/usr/bin/env python3
"""Time of the stack producers alone (HIP events on the context's stream): 27 warps of an 848x480 frame, 27 point-cloud
renders of a ~1.3 M-point plane.  python tools/producer_time.py"""
import os, sys
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
st = torch.cuda.Stream()
torch.cuda.set_stream(st)
dx, dr = torch.from_numpy(xyz).cuda(), torch.from_numpy(red).cuda()
ctx = nmi.NmiContext(w, h)
ctx.set_stream(st.cuda_stream)
Twc = np.eye(4, dtype=np.float32)
pos, look, up = Twc[:3, 3], Twc[:3, 3] + Twc[:3, 2], Twc[:3, 1]
g = H.SearchKernel.make([3] * 6, [0.2, 0.2, 0.5, 0.02, 0.02, 0.05])
cells = [(sx, sy_, sz) for sz in range(3) for sy_ in range(3) for sx in range(3)]
mvps = np.stack([capi.render_mvp(rp, pos, look, up, H.calculate_translation(Twc, g, *c)) for c in cells])
frame = torch.from_numpy(sy.camera_frame(sy.scene(w, h, 5), 6)).cuda()
rs = torch.empty((27, h, w), dtype=torch.uint8, device="cuda")
ws = torch.empty((27, h, w), dtype=torch.uint8, device="cuda")


def timed(fn, n=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(st)
    for _ in range(n):
        fn()
    e1.record(st)
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


for steps in ((0.02, 0.02, 0.05), (0.005, 0.005, 0.0125), (0.1, 0.1, 0.2)):
    homs = capi.warp_homographies(K, (3, 3, 3), steps)
    us = timed(lambda: ctx.warp_stack(frame, homs, out=ws, sync=False))
    print(f"warp stack, 27 warps {w}x{h}, steps {steps}: {us:.1f} us = {2 * w * h * 27 / us / 1e6:.2f} TB/s algorithmic (2*W*H per warp)")
us = timed(lambda: ctx.render_points(dx, dr, mvps, 3.0, out=rs, sync=False))
print(f"point render, 27 views, {xyz.shape[0]} points, size 3: {us:.1f} us (clear + splat + resolve)")
