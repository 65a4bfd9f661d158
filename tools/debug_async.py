import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import orbslam2_nmi_amd as nmi
from orbslam2_nmi_amd import synthetic as sy
wl = sy.workload(640, 480, 27, 27)
rs, ws = torch.from_numpy(wl["render_stack"]).cuda(), torch.from_numpy(wl["warp_stack"]).cuda()
ctx = nmi.NmiContext(640, 480)
st = torch.cuda.Stream(); torch.cuda.set_stream(st); ctx.set_stream(st.cuda_stream)
keys = torch.zeros(64, dtype=torch.int64, device="cuda")
for mode in ("async", "sync-each", "async-own-stream"):
    keys.zero_()
    if mode == "async-own-stream":
        ctx.set_stream(None)
    torch.cuda.synchronize()
    for i in range(20):
        ctx.search_grid_shard(rs, 0, 27, ws, key_out=keys[i:i + 1], blocking=False)
        if mode == "sync-each":
            torch.cuda.synchronize()
    if mode == "async-own-stream":
        ctx.synchronize()
    got = [nmi.key_unpack(k)[0] for k in keys[:20].cpu().tolist()]
    print(mode, got)
