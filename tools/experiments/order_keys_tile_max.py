import sys, numpy as np
sys.path.insert(0,'tools/sim')
from order_sim import makespan
from scipy import ndimage as ndi
d=np.load(sys.argv[1]); lanes=256*1024
W,H=int(d['W']),int(d['H']); GX,GY=W//32,H//32
cost=lambda S,E: 7.0*S.astype(np.float64)+64.0*E.astype(np.float64)
probe,frame=cost(d['probe_S'],d['probe_E']),cost(d['frame_S'],d['frame_E'])
main_c=np.maximum(frame-probe,0); ideal=main_c.sum()/lanes
pE=d['probe_E']; ps=max(1,int(d['spp'])//16)
tile=probe.reshape(-1,64); key=0.5*probe+0.5*np.repeat(tile.mean(1),64)
hit=(pE>ps)
def to_grid(ts): return ts.reshape(GY,GX,4,4).transpose(0,2,1,3).reshape(GY*4,GX*4)
def from_grid(g): return np.repeat(g.reshape(GY,4,GX,4).transpose(0,2,1,3).reshape(-1),64)
tmax=np.repeat(tile.max(1),64); nmax=from_grid(ndi.maximum_filter(to_grid(tile.max(1)),size=3))
m,dry=makespan(main_c,key,lanes); print("product: %.3f (dry %.3f)"%(m/ideal,dry/m))
for a in (0.25,0.5,0.75,1.0):
    for nm,tm in (("tile",tmax),("3x3 tiles",nmax)):
        k=np.where(hit,np.maximum(key,a*tm),key); m,dry=makespan(main_c,k,lanes)
        k2=np.maximum(key,a*tm); m2,dry2=makespan(main_c,k2,lanes)
        moved=(k>key).mean()
        print("hit pixels: key = max(key, %.2f x %s max): %.3f (dry %.3f) [%.3f of pixels lifted]; all pixels: %.3f"%(a,nm,m/ideal,dry/m,moved,m2/ideal))
