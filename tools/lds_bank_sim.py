"""Simulates the LDS bank conflicts of the histogram phase on the benchmark workload (CPU only): lanes on the busiest bank of a
32-lane access group per wavefront atomic, for the packed layout as it is (bank = d2 mod 32) and with every joint row rotated by a
function of its render intensity.  python tools/lds_bank_sim.py"""
import numpy as np, sys
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from orbslam2_nmi_amd import synthetic as sy
wl = sy.workload(640,480,27,27,seed=1234)
rs, ws = wl["render_stack"], wl["warp_stack"]; bu = wl["bottom_up"]
def cost(d1,d2,rot,distinct):
    # d1,d2: arrays [n_instr, 64]; word = d1*128 + ((d2&127)+rot(d1))&127 ; bank = word%32
    w = d1.astype(np.int64)*128 + (((d2&127)+rot(d1))&127)
    tot=0
    for half in (slice(0,32),slice(32,64)):
        ww = w[:,half]; b = ww%32
        n = ww.shape[0]
        if distinct:
            # count distinct addresses per bank: sort rows
            s = np.sort(ww,axis=1)
            newaddr = np.ones_like(s,dtype=bool); newaddr[:,1:] = s[:,1:]!=s[:,:-1]
            bb = s%32
            cnt = np.zeros((n,32),np.int32)
            rows = np.repeat(np.arange(n),32).reshape(n,32)
            np.add.at(cnt,(rows[newaddr],bb[newaddr]),1)
        else:
            cnt = np.zeros((n,32),np.int32)
            rows = np.repeat(np.arange(n),32).reshape(n,32)
            np.add.at(cnt,(rows,b),1)
        tot += cnt.max(axis=1).sum()
    return tot/ (2*w.shape[0])
def instrs(r,wp):
    # chunk c = it*1024 + tid; lane's 16 px; instruction (it, wave, k): lanes' pixel = chunk*16+k
    r = r.ravel(); wp = wp.ravel()
    n = r.size//16
    R = r[:n*16].reshape(n,16); W = wp[:n*16].reshape(n,16)
    # waves: 64 consecutive chunks
    nw = n//64
    R = R[:nw*64].reshape(nw,64,16).transpose(0,2,1).reshape(-1,64)
    W = W[:nw*64].reshape(nw,64,16).transpose(0,2,1).reshape(-1,64)
    return R,W
for (s,w) in [(13,13),(0,0),(5,20),(20,3),(26,13),(13,26)]:
    r = rs[s][::-1] if bu else rs[s]
    R,W = instrs(r,ws[w])
    sel = np.arange(R.shape[0])
    R,W = R[sel],W[sel]
    out=[]
    for name,rot in [("base",lambda d1:0*d1),("h1",lambda d1:(d1.astype(np.int64)>>1)),("h2",lambda d1:2*(d1.astype(np.int64)>>1)),("h3",lambda d1:3*(d1.astype(np.int64)>>1)),("h5",lambda d1:5*(d1.astype(np.int64)>>1)),("h9",lambda d1:9*(d1.astype(np.int64)>>1)),("q3",lambda d1:3*(d1.astype(np.int64)>>2))]:
        out.append((name, round(cost(R,W,rot,False),2), round(cost(R,W,rot,True),2)))
    print((s,w), out)
# random reference
rng=np.random.default_rng(1)
R=rng.integers(0,256,(20000,64)); W=rng.integers(0,256,(20000,64))
print("uniform random", round(cost(R,W,lambda d1:0*d1,False),2))
