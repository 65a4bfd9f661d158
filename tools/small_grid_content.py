import sys, numpy as np, torch
sys.path.insert(0, '.')
import orbslam2_nmi_amd as nmi
from orbslam2_nmi_amd import synthetic as sy
W, H = 640, 480
wl = sy.workload(W, H, 27, 27)
def post(a, lv):
    if lv == 0: return a
    q = 256 // lv
    return (a // q * q + q // 2).astype(np.uint8)
for lv in (0, 16, 4):
    R, Wp = torch.from_numpy(post(wl["render_stack"], lv)).cuda(), torch.from_numpy(post(wl["warp_stack"], lv)).cuda()
    for (S, Wn) in ((1, 1), (9, 1), (27, 1), (9, 9), (27, 7)):
        row = []
        for split, path in ((-1, 0), (0, 0), (0, 1)):
            with nmi.NmiContext(W, H) as ctx:
                ctx.set_profiling(True); ctx.set_option(ctx.OPT_SPLIT, split); ctx.set_option(ctx.OPT_CONTENT_PATH, path)
                t = []
                for i in range(14):
                    ctx.search_grid(R[:S], Wp[:Wn]); t.append(ctx.last_kernel_ms() * 1e3)
                row.append(f"split={split} path={path}: {np.median(t[3:]):6.1f} us")
        print(f"levels={lv or 256:3d} grid {S}x{Wn}: " + "   ".join(row), flush=True)
