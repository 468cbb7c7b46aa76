"""Tuning aid (CPU, numpy + scipy): what the tiles of k_knn_cov_tile look like under different orders of the points.

One synthetic scan after the bench's prefilter (distance 1 m, voxel 0.2 m) is cut into tiles of 64 consecutive points along
  * the 3-D Morton order of the voxel hash (0.5 m cells) — what the kernel took until round 3,
  * that order refined by an exact kd partition inside windows of 256 .. 4096 points,
  * a 3-D Hilbert curve, a 2-D Morton / Hilbert curve over x, y,
  * an exact kd partition of the whole cloud into leaves of 64 (the reference point: what compact patches would give),
and for every tile the script counts what the kernel would gather and rank: the points inside the tile box grown by its density
radius (`ntile`), inside the grown box of each of its 4 sub-tiles (`sumsub`, `maxsub` — the longest sub-list sets the trip count of
the ranking loop) and the share of tiles over the 768-entry candidate list.  Run: python benchmarks/tile_order_sim.py"""
import sys

import numpy as np

sys.path.insert(0, '.')
from lidar_graph_slam_amd import synth
from scipy.spatial import cKDTree
T_gt = synth.se3(1.0,0.1,0,0,0,np.deg2rad(2.0)); 
tgt = synth.make_pair(T_gt=T_gt, T_target=synth.se3(0,0,0,0,0,0), seed=11)[1]
p = tgt[:, :3].astype(np.float32)
d = np.linalg.norm(p,axis=1); p = p[d>1.0]
# voxel 0.2 centroids
v = np.floor(p/0.2).astype(np.int64)
key = (v[:,0]+5000)*100000000+(v[:,1]+5000)*10000+(v[:,2]+5000)
u,inv = np.unique(key, return_inverse=True)
c = np.zeros((len(u),3)); np.add.at(c,inv,p); cnt=np.bincount(inv); c/=cnt[:,None]
p=c.astype(np.float32); n=len(p); print('points',n)
cell=0.5; o=p.min(0)
b=np.clip(np.floor((p-o)/cell).astype(np.int64),0,511)
def ex(x):
    x=x&0x1FF; x=(x|(x<<16))&0x030000FF; x=(x|(x<<8))&0x0300F00F; x=(x|(x<<4))&0x030C30C3; x=(x|(x<<2))&0x09249249; return x
m=ex(b[:,0])|(ex(b[:,1])<<1)|(ex(b[:,2])<<2)
order=np.argsort(m,kind='stable'); p=p[order]
tree=cKDTree(p)
k=20
dk=tree.query(p,k=k)[0][:,-1]
rows=[]
def kd4(idx):
    def split(ix):
        e=p[ix].max(0)-p[ix].min(0); ax=np.argmax(e); s=ix[np.argsort(p[ix,ax],kind='stable')]; h=len(s)//2; return s[:h],s[h:]
    a,b_=split(idx); return [*split(a),*split(b_)]
def cheb_count(lo,hi):
    return np.count_nonzero(np.all((p>=lo)&(p<=hi),axis=1))
big=0
for t0 in range(0,n-63,64):
    idx=np.arange(t0,t0+64)
    lo=p[idx].min(0); hi=p[idx].max(0); e=np.sort(hi-lo)
    area=max(e[2]*e[1],0.16*cell*cell); r0=min(max(np.sqrt(k*area/(np.pi*64)),0.6*cell),16*cell)
    ntile=cheb_count(lo-r0,hi+r0)
    subs=kd4(idx)
    nsub=[];r0s=[]
    for sidx in subs:
        l=p[sidx].min(0); h=p[sidx].max(0); es=np.sort(h-l); a=max(es[2]*es[1],0.16*cell*cell)
        rs=min(r0, min(max(1.3*np.sqrt(k*a/(np.pi*16)),0.6*cell),16*cell)); r0s.append(rs)
        nsub.append(cheb_count(l-rs,h+rs))
    need=dk[idx].max()
    rows.append((t0,e[2],e[1],e[0],r0,ntile,sum(nsub),max(nsub),np.mean(r0s),need,np.linalg.norm(p[idx].mean(0))))
rows=np.array(rows)
print('tiles',len(rows),'tile list>768:',np.mean(rows[:,5]>768),' union(sum sub)>768:',np.mean(rows[:,6]>768), 'max sub>768', np.mean(rows[:,7]>768))
bigs=rows[rows[:,6]>768]
np.set_printoptions(precision=2,suppress=True,linewidth=200)
print('cols: t0 emax emid emin r0 ntile sumsub maxsub r0s need range')
print(bigs[:40])
print('median of normal:',np.median(rows[rows[:,5]<=768],axis=0))

print("==== kd-partition tiles")
def kd_leaves(ix, leaf):
    if len(ix) <= leaf: return [ix]
    e=p[ix].max(0)-p[ix].min(0); ax=np.argmax(e); s=ix[np.argsort(p[ix,ax],kind='stable')]
    # split at a multiple of leaf
    h=(len(s)//2 + leaf-1)//leaf*leaf
    if h>=len(s): h=len(s)//2
    return kd_leaves(s[:h],leaf)+kd_leaves(s[h:],leaf)
def stats(tiles,name):
    rows=[]
    for idx in tiles:
        lo=p[idx].min(0); hi=p[idx].max(0); e=np.sort(hi-lo)
        area=max(e[2]*e[1],0.16*cell*cell); r0=min(max(np.sqrt(k*area/(np.pi*len(idx))),0.6*cell),16*cell)
        ntile=cheb_count(lo-r0,hi+r0)
        subs=kd4(idx) if len(idx)>=4 else [idx]
        nsub=[]
        for sidx in subs:
            l=p[sidx].min(0); h=p[sidx].max(0); es=np.sort(h-l); a=max(es[2]*es[1],0.16*cell*cell)
            rs=min(r0, min(max(1.3*np.sqrt(k*a/(np.pi*len(sidx))),0.6*cell),16*cell))
            nsub.append(cheb_count(l-rs,h+rs))
        need=dk[idx].max()
        # with the true needed radius per sub-tile
        nsub_need=[]
        for sidx in subs:
            l=p[sidx].min(0); h=p[sidx].max(0); rn=dk[sidx].max(); nsub_need.append(cheb_count(l-rn,h+rn))
        rows.append((e[2],e[1],e[0],r0,ntile,sum(nsub),max(nsub),np.mean(nsub),need,np.mean(nsub_need),max(nsub_need)))
    rows=np.array(rows)
    print(name,'tiles',len(rows),'ntile>768:',np.mean(rows[:,4]>768),' sumsub>768:',np.mean(rows[:,5]>768),'maxsub>768',np.mean(rows[:,6]>768))
    print(' mean: emax %.2f emid %.2f emin %.2f r0 %.2f ntile %.0f sumsub %.0f maxsub %.0f meansub %.0f need %.2f | with true radius: meansub %.0f maxsub %.0f'%tuple(rows.mean(0)))
    print(' median:',np.median(rows,axis=0))
stats([np.arange(t0,t0+64) for t0 in range(0,n-63,64)],'morton')
stats(kd_leaves(np.arange(n),64),'kd')
for W in (256,512,1024,2048,4096):
    tiles=[]
    for w0 in range(0,n,W):
        ix=np.arange(w0,min(w0+W,n))
        tiles+= [t for t in kd_leaves(ix,64) if len(t)>=16]
    stats(tiles,'window %d'%W)
def hilbert_index(X, bits):
    # Skilling's transpose algorithm, vectorised; X: [n,3] ints
    X = X.copy().astype(np.int64); nD=3
    M = 1 << (bits-1)
    Q = M
    while Q > 1:
        P = Q-1
        for i in range(nD):
            sel = (X[:,i] & Q)!=0
            X[sel,0] ^= P
            t = (X[:,0]^X[:,i]) & P
            t[sel]=0
            X[:,0]^=t; X[:,i]^=t
        Q >>= 1
    for i in range(1,nD): X[:,i]^=X[:,i-1]
    t=np.zeros(len(X),dtype=np.int64)
    Q=M
    while Q>1:
        sel=(X[:,nD-1]&Q)!=0
        t[sel]^=Q-1
        Q>>=1
    for i in range(nD): X[:,i]^=t
    # interleave
    h=np.zeros(len(X),dtype=np.int64)
    for b in range(bits-1,-1,-1):
        for i in range(nD):
            h=(h<<1)|((X[:,i]>>b)&1)
    return h
for cs,bits in ((0.5,9),(0.125,11)):
    bb_=np.clip(np.floor((p-o)/cs).astype(np.int64),0,(1<<bits)-1)
    h=hilbert_index(bb_,bits)
    oh=np.argsort(h,kind='stable')
    stats([oh[t0:t0+64] for t0 in range(0,n-63,64)],'hilbert cell %.3f'%cs)
    mm=np.zeros(n,dtype=np.int64)
    for b in range(bits-1,-1,-1):
        for i in (2,1,0):
            mm=(mm<<1)|((bb_[:,i]>>b)&1)
    om=np.argsort(mm,kind='stable')
    stats([om[t0:t0+64] for t0 in range(0,n-63,64)],'morton cell %.3f'%cs)
print("==== hilbert + window kd")
for W in (256,1024):
    tiles=[]
    for w0 in range(0,n,W):
        ix=oh[w0:min(w0+W,n)]
        tiles+= [t for t in kd_leaves(ix,64) if len(t)>=16]
    stats(tiles,'hilbert window %d'%W)
print("==== other static keys")
def morton2(ix,iy,bits):
    mm=np.zeros(len(ix),dtype=np.int64)
    for b in range(bits-1,-1,-1):
        mm=(mm<<1)|((iy>>b)&1); mm=(mm<<1)|((ix>>b)&1)
    return mm
def hilbert2(ix,iy,bits):
    x=ix.copy(); y=iy.copy(); d=np.zeros(len(x),dtype=np.int64)
    s=1<<(bits-1)
    while s>0:
        rx=((x&s)>0).astype(np.int64); ry=((y&s)>0).astype(np.int64)
        d+=s*s*((3*rx)^ry)
        # rotate
        sel=(ry==0)
        flip=sel&(rx==1)
        x=np.where(flip,s-1-x,x); y=np.where(flip,s-1-y,y)
        x2=np.where(sel,y,x); y2=np.where(sel,x,y); x,y=x2,y2
        x&=s-1; y&=s-1  # keep low bits
        s>>=1
    return d
for cs in (0.1,0.25):
    q=np.floor((p-o)/cs).astype(np.int64)
    om=np.argsort(morton2(q[:,0],q[:,1],11),kind='stable')
    stats([om[t0:t0+64] for t0 in range(0,n-63,64)],'morton2D xy cell %.2f'%cs)
    ohh=np.argsort(hilbert2(q[:,0],q[:,1],11),kind='stable')
    stats([ohh[t0:t0+64] for t0 in range(0,n-63,64)],'hilbert2D xy cell %.2f'%cs)
