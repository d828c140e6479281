#!/usr/bin/env python3
"""How many of the 125 stencil cells a 64-lane wave of the tile kernels has to evaluate (any lane needs it) when the
records of a tile are ordered by sub-cell bins of nbx x nby x nbz (CPU simulation)."""
import numpy as np
rng=np.random.default_rng(1)
# candidate cells: offsets a,b,c in -2..2; hull81: those with min distance <= 2 from home cell box
offs=[(a,b,c) for a in range(-2,3) for b in range(-2,3) for c in range(-2,3)]
def need(u):  # u: (n,3) sub-cell offsets in [-.5,.5); returns (n,125) bool: q<=2 with h=d
    o=np.array(offs,float)
    d=u[:,None,:]-o[None,:,:]
    return (d**2).sum(-1)<=4.0
def sim(nbx,nby,nbz,ntile=200,npart=1024):
    tot=0;waves=0;per=0
    for t in range(ntile):
        n=rng.poisson(npart)
        u=rng.random((n,3))-0.5
        bx=np.minimum((u[:,0]+.5)*nbx,nbx-1).astype(int); by=np.minimum((u[:,1]+.5)*nby,nby-1).astype(int); bz=np.minimum((u[:,2]+.5)*nbz,nbz-1).astype(int)
        key=(bx*nby+by)*nbz+bz
        order=np.argsort(key,kind='stable')
        nd=need(u[order])
        per+=nd.sum()
        # work item chunks of 2048, 256 threads: pass p covers records [256p,256p+256): wave w lanes 64w..64w+63
        for s in range(0,n,64):
            tot+=nd[s:s+64].any(0).sum(); waves+=1
    return tot/waves, per/ (ntile*npart)
for b in [(1,1,1),(2,2,2),(2,2,4),(2,4,4),(4,4,4),(4,4,8)]:
    print(b, "cands/wave %.1f  need/particle %.1f"%sim(*b))
print("npart 2048")
for b in [(2,2,2),(2,2,4),(2,4,4)]:
    print(b, "cands/wave %.1f  need/particle %.1f"%sim(*b,ntile=100,npart=2048))
print("npart 4096")
for b in [(2,2,2),(2,2,4),(2,4,4),(4,4,4)]:
    print(b, "cands/wave %.1f  need/particle %.1f"%sim(*b,ntile=50,npart=4096))
