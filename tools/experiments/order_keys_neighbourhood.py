import sys, numpy as np
sys.path.insert(0,'tools/sim')
from order_sim import makespan
from scipy import ndimage as ndi
d=np.load(sys.argv[1]); lanes=256*1024
W,H=int(d['W']),int(d['H'])
cost=lambda S,E: 7.0*S.astype(np.float64)+64.0*E.astype(np.float64)
probe,frame=cost(d['probe_S'],d['probe_E']),cost(d['frame_S'],d['frame_E'])
main_c=np.maximum(frame-probe,0); n=len(main_c); ideal=main_c.sum()/lanes
GX,GY=W//32,H//32
def to_grid(tile_stat):   # tile_stat: (groups*16,) -> (GY*4, GX*4)
    t=tile_stat.reshape(GY,GX,4,4)       # [gy,gx,ty,tx]
    return t.transpose(0,2,1,3).reshape(GY*4,GX*4)
def from_grid(g):
    t=g.reshape(GY,4,GX,4).transpose(0,2,1,3).reshape(-1)
    return np.repeat(t,64)
tile=probe.reshape(-1,64)
tmean_g=to_grid(tile.mean(1)); tmax_g=to_grid(tile.max(1))
keys={}
keys['product']=0.5*probe+0.5*np.repeat(tile.mean(1),64)
keys['tile max']=np.repeat(tile.max(1),64)
for r in (1,2,4):
    keys['nbhd%d max'%r]=from_grid(ndi.maximum_filter(tmax_g,size=2*r+1))
    keys['nbhd%d mean'%r]=from_grid(ndi.uniform_filter(tmean_g,size=2*r+1))
    keys['probe+nbhd%d max'%r]=probe+from_grid(ndi.maximum_filter(tmax_g,size=2*r+1))
    keys['0.5 probe + 0.5 nbhd%d mean'%r]=0.5*probe+0.5*from_grid(ndi.uniform_filter(tmean_g,size=2*r+1))
# true-cost spatial smoothness: how well would the TRUE tile mean predict?
tt=main_c.reshape(-1,64).mean(1); keys['(oracle) true tile mean']=np.repeat(tt,64)
keys['(oracle) true nbhd2 mean']=from_grid(ndi.uniform_filter(to_grid(tt),size=5))
for k,v in keys.items():
    m,dry=makespan(main_c,v,lanes); print("  %-34s makespan %.3f x ideal, dry at %.3f"%(k,m/ideal,dry/m))
# correlation
print("corr(probe, main) %.3f  corr(tile mean probe, main) %.3f  corr(true tile mean, main) %.3f"%(np.corrcoef(probe,main_c)[0,1], np.corrcoef(np.repeat(tile.mean(1),64),main_c)[0,1], np.corrcoef(np.repeat(tt,64),main_c)[0,1]))
