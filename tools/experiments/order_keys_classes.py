import sys, numpy as np
sys.path.insert(0,'tools/sim')
from order_sim import makespan
d=np.load(sys.argv[1]); lanes=256*1024
cost=lambda S,E: 7.0*S.astype(np.float64)+64.0*E.astype(np.float64)
probe,frame=cost(d['probe_S'],d['probe_E']),cost(d['frame_S'],d['frame_E'])
main_c=np.maximum(frame-probe,0); ideal=main_c.sum()/lanes
pE=d['probe_E']; spp=int(d['spp']); ps=max(1,spp//16)
tile=probe.reshape(-1,64); key=0.5*probe+0.5*np.repeat(tile.mean(1),64)
hit=(pE>ps)
m,dry=makespan(main_c,key,lanes); print("product: %.3f (dry %.3f)"%(m/ideal,dry/m))
m,dry=makespan(main_c,key+1e9*hit,lanes); print("pixels with a hit in the probe first: %.3f (dry %.3f)  [%.3f of pixels]"%(m/ideal,dry/m,hit.mean()))
th=np.repeat(hit.reshape(-1,64).any(1),64)
m,dry=makespan(main_c,key+1e9*th,lanes); print("pixels of tiles with a hit first: %.3f (dry %.3f)  [%.3f of pixels]"%(m/ideal,dry/m,th.mean()))
# quantised key as the kernel would build it: 16 bins per octave
def q(c): 
    c=np.maximum(c,1); e=np.floor(np.log2(c)); fr=np.floor((c/2**e-1)*16); return e*16+fr
m,dry=makespan(main_c,q(key)+1e6*hit,lanes); print("  same (pixel flag), key quantised to 16 bins per octave: %.3f"%(m/ideal))
# number of hits as the class (more hits = riskier?)
m,dry=makespan(main_c,key+1e9*np.minimum(pE-ps,3),lanes); print("classes by number of hits in the probe (0,1,2,3+): %.3f"%(m/ideal))
