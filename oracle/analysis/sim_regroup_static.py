"""TEST INFRASTRUCTURE (part of oracle/): offline scheduling study on traced rays (see sched_sim.py)."""
import sys, numpy as np
import os; sys.path.insert(0, os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import sched_sim as SS
class A: width=1920; height=1080; bounces=4; detail=1.0; obj=None
tr = SS.trace_frame(A)
REC=SS.REC
rng=np.random.default_rng(3)
tiles_x, tiles_y = 1920//16, 1080//16
sel = rng.choice(tiles_x*(tiles_y), size=300, replace=False)
def cost(n, lc):  # n: [64] counts for one bounce, lc [64,REC-2]
    m=int(n.max()); 
    if m==0: return 0,0,0
    steps=min(m,REC-2); idx=np.arange(steps)[None,:]; act=idx<n[:,None]
    l=np.where(act, lc[:,:steps],0).max(axis=0)
    return m, int(l.sum()), int((n>0).sum())
tot={k:[0,0,0] for k in ('base','perfect','prev','random')}
perb={b:[0,0,0] for b in range(4)}
for s in sel:
    ty,tx=divmod(int(s),tiles_x)
    t=tr[ty*16:ty*16+16, tx*16:tx*16+16].reshape(16,16,8,REC)
    # baseline waves = 8x8 quadrants
    quads=[t[0:8,0:8],t[0:8,8:16],t[8:16,0:8],t[8:16,8:16]]
    flat=np.concatenate([q.reshape(64,8,REC) for q in quads])  # 256 rays, wave-major
    n=flat[:,:,0].astype(np.int32)|(flat[:,:,1].astype(np.int32)<<8)
    lc=flat[:,:,2:]
    for b in range(4):
        for name in tot:
            if name=='base' or b==0: order=np.arange(256)
            elif name=='perfect': order=np.argsort(-n[:,b],kind='stable')
            elif name=='prev': order=np.argsort(-n[:,b-1],kind='stable')
            else: order=rng.permutation(256)
            for w in range(4):
                idx=order[w*64:(w+1)*64]
                m,l,r=cost(n[idx,b], lc[idx,b])
                tot[name][0]+=m; tot[name][1]+=l; tot[name][2]+=r
                if name=='base':
                    perb[b][0]+=m; perb[b][1]+=l; perb[b][2]+=r
for name,(m,l,r) in tot.items():
    print(name,'node steps/ray %.3f leaf steps/ray %.3f  instr/ray %.1f'%(m/r,l/r,(m*185+l*85)/r))
for b,(m,l,r) in perb.items():
    print('bounce',b,'node steps/ray %.3f leaf %.3f instr/ray %.1f'%(m/r,l/r,(m*185+l*85)/r))
