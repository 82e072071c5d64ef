import numpy as np
t=np.load('gpurun_out/r04w/times_c5.npz'); c=np.load('gpurun_out/r04s/costs_c5.npz')
cost=(t['fresh_cost']&0x7FFFFFFF).astype(np.float64); t0=t['fresh_t0']/100.0; t1=t['fresh_t1']/100.0; span=t1.max()
pS,pE,fS,fE=[c[k].astype(np.float64) for k in ('probe_S','probe_E','frame_S','frame_E')]
probe=7*pS+64*pE
tile=probe.reshape(-1,64); tm=np.repeat(tile.mean(1),64); tmax=np.repeat(tile.max(1),64)
key=0.5*probe+0.5*tm
late=(t1>0.85*span)
print("pixels ending after 85%% of the span: %d"%late.sum())
print(" probe steps/sample pct 10/50/90:",np.percentile(pS[late]/4,[10,50,90]).round(1)," events/sample:",np.percentile(pE[late]/4,[10,50,90]).round(2))
print(" key / max key pct 10/50/90:",(np.percentile(key[late],[10,50,90])/key.max()).round(3)," tile max / max key:",(np.percentile(tmax[late],[10,50,90])/key.max()).round(3))
print(" main steps/sample pct:",np.percentile((fS-pS)[late]/60,[10,50,90]).round(0)," main events/sample",np.percentile((fE-pE)[late]/60,[10,50,90]).round(2))
# population with similar probe signature
sig=(pE/4>=1.5)&(pS/4<60)
print("pixels with >= 1.5 events/sample and < 60 steps/sample in the probe: %.3f of all; share of them that are late: %.4f; late covered: %.3f"%(sig.mean(),(late&sig).sum()/sig.sum(),(late&sig).sum()/late.sum()))
# probability of becoming heavy (main steps/sample > 200) as a function of probe events/sample among cheap-probe pixels
cheap=(pS/4<60)
heavy=((fS-pS)/60>200)
for lo,hi in ((0.9,1.1),(1.1,1.6),(1.6,2.1),(2.1,3.1),(3.1,99)):
    m=cheap&(pE/4>=lo)&(pE/4<hi)
    print("   cheap probe, events/sample in [%.1f,%.1f): %.3f of pixels; become heavy: %.4f"%(lo,hi,m.mean(),heavy[m].mean() if m.sum() else 0))
# spatial: are late pixels clustered in tiles with heavy neighbours (true)?
ht=np.repeat(heavy.reshape(-1,64).mean(1),64)
print(" share of heavy pixels in the late pixels' own tile (true, main): pct 10/50/90",np.percentile(ht[late],[10,50,90]).round(3))
pheavy=(pS/4>200); pht=np.repeat(pheavy.reshape(-1,64).mean(1),64)
print(" share of pixels ALREADY heavy in the probe in their tile: pct 10/50/90",np.percentile(pht[late],[10,50,90]).round(3), " (all cheap-probe hit pixels:",np.percentile(pht[cheap&(pE/4>1)],[10,50,90]).round(3),")")
