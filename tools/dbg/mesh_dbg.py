import sys, os, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import orbslam2_nmi_amd as nmi
from orbslam2_nmi_amd import capi
from oracle import mesh_oracle_np as mo
import test_render as tr
w, h = 320, 200
gx, gu, rgb, rp = tr.ground_mesh(w, h)
gx, gu = tr._both_windings(gx, gu)
px, pu, _, _ = tr.plane_mesh(w, h, depth=30.0, nx=6, ny=4)
xyz, uv = np.concatenate([gx, px]), np.concatenate([gu, pu])
mvps = np.stack([capi.render_mvp(rp, (0, 0, 0), (0, 0, 1), (0, -1, 0), t)
                 for t in ((0, 0, 0), (0.4, -0.3, 1.0), (-0.8, 0.2, -3.0), (2.5, 0.5, 0), (0, -1.0, 2.0))])
exp = mo.render_stack(xyz, uv, mo.mip_luma(rgb), mvps, w, h)
# which triangle wins each pixel in the twin: render each triangle alone?  cheaper: per-group renders
with nmi.NmiContext(w, h) as ctx, nmi.NmiTexture(ctx, rgb) as tex:
    got = ctx.render_mesh(torch.from_numpy(xyz).cuda(), torch.from_numpy(uv).cuda(), tex, mvps).cpu().numpy()
    got_g = ctx.render_mesh(torch.from_numpy(gx).cuda(), torch.from_numpy(gu).cuda(), tex, mvps).cpu().numpy()
    got_p = ctx.render_mesh(torch.from_numpy(px).cuda(), torch.from_numpy(pu).cuda(), tex, mvps).cpu().numpy()
exp_g = mo.render_stack(gx, gu, mo.mip_luma(rgb), mvps, w, h)
exp_p = mo.render_stack(px, pu, mo.mip_luma(rgb), mvps, w, h)
for name, a, b in (("all", got, exp), ("ground", got_g, exp_g), ("plane", got_p, exp_p)):
    d = np.abs(a.astype(int) - b.astype(int))
    print(name, "per view max", d.reshape(5, -1).max(1), "frac", (d != 0).reshape(5, -1).mean(1), "cov", ((a == 255) == (b == 255)).all(), flush=True)
np.savez_compressed(os.path.join(ROOT, "gpurun_out", "mesh_dbg.npz"), got=got, exp=exp, got_g=got_g, exp_g=exp_g, got_p=got_p, exp_p=exp_p)
