"""TEST INFRASTRUCTURE (part of oracle/): offline scheduling study on traced rays (see sched_sim.py)."""
import sys, numpy as np
import os; sys.path.insert(0, os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import sched_sim as SS
class A: width=1920; height=1080; bounces=4; detail=1.0; obj=None
tr = SS.trace_frame(A)
REC=SS.REC
rng=np.random.default_rng(3)
tiles_x, tiles_y = 1920//16, 1080//16
sel = rng.choice(tiles_x*(tiles_y), size=200, replace=False)
CN,CL,CR=185,85,90
def run_block(n, lc, K, sortkey):
    # n [256], lc [256, REC-2]; returns node_steps, leaf_steps, regroups(wave-level)
    alive=np.nonzero(n>0)[0]
    step=np.zeros(256,np.int32)
    order=np.arange(256)  # ray ids in lane order; dead rays removed at regroup
    rays=[i for i in range(256) if n[i]>0] if K else None
    ns=ls=rg=0
    if not K:
        for w in range(4):
            idx=np.arange(w*64,(w+1)*64)
            m=int(n[idx].max())
            if m==0: continue
            st=min(m,REC-2); ar=np.arange(st)[None,:]; act=ar<n[idx][:,None]
            ns+=m; ls+=int(np.where(act,lc[idx,:st],0).max(axis=0).sum())
        return ns,ls,0
    cur=np.array(rays,dtype=np.int64)
    first=True
    while cur.size:
        nw=(cur.size+63)//64
        for w in range(nw):
            idx=cur[w*64:(w+1)*64]
            rem=n[idx]-step[idx]
            m=int(min(K,rem.max()))
            for j in range(m):
                act=(step[idx]+j)<n[idx]
                pos=np.minimum(step[idx]+j,REC-3)
                l=np.where(act, lc[idx,pos],0).max()
                ls+=int(l)
            ns+=m
            step[idx]=np.minimum(step[idx]+m, n[idx])
        rg+=nw
        cur=cur[step[cur]<n[cur]]
        if sortkey=='depth' and cur.size:
            pass
    return ns,ls,rg
res={}
for K in (0,4,8,16,32):
    ns=ls=rg=r=0
    for s in sel:
        ty,tx=divmod(int(s),tiles_x)
        t=tr[ty*16:ty*16+16, tx*16:tx*16+16]
        quads=[t[0:8,0:8],t[0:8,8:16],t[8:16,0:8],t[8:16,8:16]]
        flat=np.concatenate([q.reshape(64,8,REC) for q in quads])
        n=flat[:,:,0].astype(np.int32)|(flat[:,:,1].astype(np.int32)<<8)
        lc=flat[:,:,2:]
        for b in range(4):
            a,c,d=run_block(n[:,b],lc[:,b],K,None)
            ns+=a; ls+=c; rg+=d; r+=int((n[:,b]>0).sum())
    print('K',K,'node steps/ray %.3f leaf %.3f regroups/ray %.4f instr/ray %.1f'%(ns/r,ls/r,rg/r,(ns*CN+ls*CL+rg*CR)/r))
