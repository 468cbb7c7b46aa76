"""Tuning aid (CPU only, NumPy + SciPy): how many candidates a lane of a 64-point Morton tile has to rank under the cell-brick walks of
DESIGN.md 4.4 - lockstep over relative / absolute rows, per-lane independent iteration, per-lane radii - on the bench cloud."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from scipy.spatial import cKDTree
from lidar_graph_slam_amd import synth
from oracle import oracle_py as O
src,tgt,T=synth.make_pair()
s=O.voxel_grid(O.distance_filter(src,1.0),0.2)[0][:,:3].astype(np.float64)
# morton order of 0.5 m cells
lo=s.min(0); c=np.floor((s-lo)/0.5).astype(np.int64)
def part(x):
    x&=0x1FF; x=(x|(x<<16))&0x030000FF; x=(x|(x<<8))&0x0300F00F; x=(x|(x<<4))&0x030C30C3; x=(x|(x<<2))&0x09249249; return x
key=part(c[:,0].copy())|(part(c[:,1].copy())<<1)|(part(c[:,2].copy())<<2)
order=np.argsort(key,kind='stable'); s=s[order]
tree=cKDTree(s)
rk=tree.query(s,k=21)[0][:,20]
k=20
rng=np.random.default_rng(0)
tiles=rng.choice(len(s)//64, 150, replace=False)
res={}
def run(name, hscale, hyz_scale, rmult):
    tot_lock=tot_flat=tot_mean=0; rows_tot=0; fails=0; nl=0
    for t in tiles:
        q=s[t*64:(t+1)*64]
        ext=q.max(0)-q.min(0); e=np.sort(ext); area=max(e[2]*e[1],0.16*0.25)
        r0=np.clip(np.sqrt(k*area/(np.pi*64)),0.3,8.0)
        R=rmult*r0
        hx=hscale*r0; hyz=hyz_scale*r0
        lo=q.min(0)-R; hi=q.max(0)+R
        idx=tree.query_ball_point((lo+hi)/2, np.linalg.norm(hi-lo)/2+1e-6)
        cand=s[idx]; cand=cand[np.all((cand>=lo)&(cand<=hi),axis=1)]
        h=np.array([hx,hyz,hyz])
        cc=np.floor((cand-lo)/h).astype(int)
        # per lane
        counts=[]  # list per lane of dict row->count
        allrows=set()
        lanes=[]
        for j in range(64):
            c0=np.floor((q[j]-R-lo)/h).astype(int); c1=np.floor((q[j]+R-lo)/h).astype(int)
            m=np.all((cc>=c0)&(cc<=c1),axis=1)
            sub=cc[m]
            wy=c1[1]-c0[1]+1; wz=c1[2]-c0[2]+1
            rc=np.zeros((wy,wz),int)
            np.add.at(rc,(sub[:,1]-c0[1],sub[:,2]-c0[2]),1)
            lanes.append(rc)
            fails+= rk[t*64+j]>R*0.999; nl+=1
        # lockstep relative rows: pad to max dims
        my=max(r.shape[0] for r in lanes); mz=max(r.shape[1] for r in lanes)
        P=np.zeros((64,my,mz),int)
        for j,r in enumerate(lanes): P[j,:r.shape[0],:r.shape[1]]=r
        tot_lock+=P.max(0).sum(); tot_flat+=P.sum((1,2)).max(); tot_mean+=P.sum((1,2)).mean(); rows_tot+=my*mz
    n=len(tiles)
    print(f"{name:30s} rows {rows_tot/n:5.1f} lockstep {tot_lock/n:6.1f} flat(max lane) {tot_flat/n:6.1f} mean/lane {tot_mean/n:6.1f} fail {fails/nl:.2f}")
run("h=.6 iso R=1.0", .6,.6,1.0)
run("h=.6 iso R=1.3", .6,.6,1.3)
run("h=.6 iso R=1.5", .6,.6,1.5)
run("hx=.3 hyz=1.0 R=1.0", .3,1.0,1.0)
run("hx=.3 hyz=1.3 R=1.3", .3,1.3,1.3)
run("hx=.3 hyz=.65 R=1.3", .3,.65,1.3)
print("---- per-lane adaptive (oracle knowledge of r_k): candidates in the cell window of cube radius R_l")
def run2(name, hscale, levels):
    tot_flat=tot_mean=0; nl=0; tot_lock=0
    for t in tiles:
        q=s[t*64:(t+1)*64]
        ext=q.max(0)-q.min(0); e=np.sort(ext); area=max(e[2]*e[1],0.16*0.25)
        r0=np.clip(np.sqrt(k*area/(np.pi*64)),0.3,8.0)
        h=hscale*r0
        Rmax=max(levels)*r0*1.01+rk[t*64:(t+1)*64].max()
        lo=q.min(0)-Rmax; hi=q.max(0)+Rmax
        idx=tree.query_ball_point((lo+hi)/2, np.linalg.norm(hi-lo)/2+1e-6)
        cand=s[idx]; cand=cand[np.all((cand>=lo)&(cand<=hi),axis=1)]
        cc=np.floor((cand-lo)/h).astype(int)
        per=[]
        for j in range(64):
            r=rk[t*64+j]
            Rl=next((m*r0 for m in levels if m*r0>=r*1.001), r*1.05)
            c0=np.floor((q[j]-Rl-lo)/h).astype(int); c1=np.floor((q[j]+Rl-lo)/h).astype(int)
            per.append(int(np.all((cc>=c0)&(cc<=c1),axis=1).sum()))
        per=np.array(per); tot_flat+=per.max(); tot_mean+=per.mean()
    n=len(tiles)
    print(f"{name:40s} flat(max lane) {tot_flat/n:6.1f} mean/lane {tot_mean/n:6.1f}")
run2("h=.3r0 levels .4,.55,.7,.85,1,1.3,1.7,2.2", .3, [.4,.55,.7,.85,1,1.3,1.7,2.2,3])
run2("h=.2r0 same levels", .2, [.4,.55,.7,.85,1,1.3,1.7,2.2,3])
run2("h=.3r0 levels .5,.7,1,1.4,2", .3, [.5,.7,1,1.4,2,3])
run2("h=.15r0 fine levels", .15, [.3,.4,.5,.6,.7,.8,.9,1,1.15,1.3,1.5,1.7,2,2.5,3])
print("---- absolute-row lockstep")
def run3(name, hscale, rmult, axis_order=(0,1,2)):
    tot_lock=tot_rows=tot_mean=0
    for t in tiles:
        q=s[t*64:(t+1)*64][:,axis_order]
        ext=q.max(0)-q.min(0); e=np.sort(ext); area=max(e[2]*e[1],0.16*0.25)
        r0=np.clip(np.sqrt(k*area/(np.pi*64)),0.3,8.0)
        R=rmult*r0; h=hscale*r0
        lo=q.min(0)-R; hi=q.max(0)+R
        sa=s[:,axis_order]
        idx=tree.query_ball_point(((lo+hi)/2)[np.argsort(axis_order)], np.linalg.norm(hi-lo)/2+1e-6)
        cand=sa[idx]; cand=cand[np.all((cand>=lo)&(cand<=hi),axis=1)]
        cc=np.floor((cand-lo)/h).astype(int)
        dims=np.floor((hi-lo)/h).astype(int)+1
        P=np.zeros((64,dims[1],dims[2]),int)
        for j in range(64):
            c0=np.floor((q[j]-R-lo)/h).astype(int); c1=np.floor((q[j]+R-lo)/h).astype(int)
            m=np.all((cc>=c0)&(cc<=c1),axis=1); sub=cc[m]
            np.add.at(P[j],(sub[:,1],sub[:,2]),1)
        mx=P.max(0)
        tot_lock+=mx.sum(); tot_rows+=(mx>0).sum(); tot_mean+=P.sum((1,2)).mean()
    n=len(tiles)
    print(f"{name:40s} nonempty rows {tot_rows/n:5.1f} lockstep steps {tot_lock/n:6.1f} mean/lane {tot_mean/n:6.1f}")
run3("abs rows h=.6 R=1.0", .6, 1.0)
run3("abs rows h=.4 R=1.0", .4, 1.0)
run3("abs rows h=.6 R=1.0 rows along z", .6, 1.0, (2,0,1))
run3("abs rows h=.6 R=1.0 rows along y", .6, 1.0, (1,2,0))
