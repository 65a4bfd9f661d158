"""Counts, on the benchmark workload (CPU only), the lanes of a wavefront atomic that land on the SAME bin as another lane of their
32-lane group, by where they come from: render background over frame border (255, 0), border only, background only, the rest.
python tools/lds_duplicate_sim.py"""
import numpy as np, sys
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from orbslam2_nmi_amd import synthetic as sy
wl = sy.workload(640,480,27,27,seed=1234)
rs, ws = wl["render_stack"], wl["warp_stack"]; bu = wl["bottom_up"]
def instrs(r,wp):
    r = r.ravel(); wp = wp.ravel()
    n = r.size//16
    R = r[:n*16].reshape(n,16); W = wp[:n*16].reshape(n,16)
    nw = n//64
    R = R[:nw*64].reshape(nw,64,16).transpose(0,2,1).reshape(-1,64)
    W = W[:nw*64].reshape(nw,64,16).transpose(0,2,1).reshape(-1,64)
    return R,W
rng=np.random.default_rng(5)
tot=np.zeros(4); n=0
for _ in range(80):
    s,w = rng.integers(0,27,2)
    r = rs[s][::-1] if bu else rs[s]
    R,W = instrs(r,ws[w])
    key = R.astype(np.int64)*256+W
    cat=np.zeros(4)
    for g0 in (0,32):
        k = np.sort(key[:,g0:g0+32],axis=1)
        same = (k[:,1:]==k[:,:-1])
        kk = k[:,1:][same]
        d1 = kk>>8; d2 = kk&255
        cat[0]+=((d1==255)&(d2==0)).sum(); cat[1]+=((d2==0)&(d1!=255)).sum(); cat[2]+=((d1==255)&(d2!=0)).sum(); cat[3]+=((d1!=255)&(d2!=0)).sum()
    tot+=cat/key.shape[0]; n+=1
print("mean duplicate lanes per wavefront instruction over 80 random candidates: (255,0) %.3f | d2==0 other %.3f | d1==255 other %.3f | rest %.3f | all %.3f" % (*(tot/n), tot.sum()/n))
