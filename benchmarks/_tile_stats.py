import ctypes, numpy as np, sys, os
sys.path.insert(0, os.getcwd())
import bench, torch
from lidar_graph_slam_amd import _lib
L = _lib.lib()
sys.argv = ['bench.py','--steps','1','--warmup','0','--pairs','64','--no-cpu-baseline']
out = (ctypes.c_ulonglong*16)()
L.mi355_debug_tile_stats(out)
buf = np.zeros((1<<18)*4, dtype=np.uint64)
L.mi355_debug_tile_log.restype = ctypes.c_uint
L.mi355_debug_tile_log(buf.ctypes.data_as(ctypes.POINTER(ctypes.c_ulonglong)), 1<<18)
bench.main()
L.mi355_debug_tile_stats(out)
v=list(out); print('tiles',v[0],'cand/tile',v[1]/max(v[0],1),'rounds/tile',v[2]/max(v[0],1),'segs/tile',v[3]/max(v[0],1),'maxcand',v[4],'fallback lanes',v[8], 'frac', v[8]/(64*max(v[0],1)))
print('fb causes: over-budget round0', v[10], 'over-budget later', v[11], '(of which cells not ok', v[15], ') rounds>=4', v[12], 'no growth', v[13], 'ambiguous', v[14])
nh = int(v[9])
n = L.mi355_debug_tile_log(buf.ctypes.data_as(ctypes.POINTER(ctypes.c_ulonglong)), 1<<18)
h = buf[4*((1<<18)-nh):4*(1<<18)].reshape(nh,4).astype(np.int64)
print('hard queries', nh)
if nh:
    hd=(h[:,1]-h[:,0])/100.0; print(' hard dur mean', hd.mean(), 'max', hd.max(), 'scanned mean', (h[:,3]>>32).mean(), 'rounds mean', (h[:,3]&255).mean(), 'whole frac', ((h[:,3]>>8)&1).mean()); cl=h[:,2]>>32; print(' per-cloud max count', np.bincount(cl).max(), 'max t', (h[:,2]&0xffffffff).max()); top=np.argsort(-hd)[:6]; print(' slowest', [(float(hd[t]), int(h[t,3]>>32), int(h[t,3]&0xffffffff)) for t in top])
a = buf[:4*n].reshape(n,4).astype(np.int64)
print('log entries', n)
# split into launches by time gaps: sort by start
o = np.argsort(a[:,0]); a = a[o]
t0 = a[:,0].min()
st = (a[:,0]-t0)/100.0; en = (a[:,1]-t0)/100.0   # us (100 MHz)
# find launches: gaps > 200us between consecutive starts
gaps = np.where(np.diff(st) > 300)[0]
bounds = [0]+list(gaps+1)+[n]
for b0,b1 in zip(bounds[:-1],bounds[1:]):
    s,e = st[b0:b1], en[b0:b1]; d = e-s
    na = a[b0:b1,3]>>32
    nr = a[b0:b1,3]&0xffffffff
    top = np.argsort(-d)[:8]
    ph = a[b0:b1,2]; print('  slowest (dur, cand, rounds, cells_us, gather_us, rank_us):', [(round(float(d[t]),0), int(na[t]), int(nr[t]), int(ph[t]>>42)/100, int((ph[t]>>21)&0x1fffff)/100, int(ph[t]&0x1fffff)/100) for t in top]); print('  mean phases us: cells', (ph>>42).mean()/100, 'gather', ((ph>>21)&0x1fffff).mean()/100, 'rank', (ph&0x1fffff).mean()/100)
    print(f'launch: waves {b1-b0} span {e.max()-s.min():.0f}us  last start {s.max()-s.min():.0f}us  dur mean {d.mean():.1f} p50 {np.median(d):.1f} p99 {np.percentile(d,99):.1f} max {d.max():.1f}  sum_dur/span {d.sum()/(e.max()-s.min()):.0f} waves avg; corr(n_a,dur)={np.corrcoef(na,d)[0,1]:.2f} us/cand {d.sum()/max(na.sum(),1):.3f}')
    # concurrency timeline
    T = np.linspace(s.min(), e.max(), 11)
    print('  resident waves at deciles:', [int(((s<=t)&(e>t)).sum()) for t in T])
