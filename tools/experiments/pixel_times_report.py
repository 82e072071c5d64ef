import sys, numpy as np
d=np.load(sys.argv[1])
for mode in ("fresh","replay"):
    c=d[mode+"_cost"]; t0=d[mode+"_t0"].astype(np.float64)/100; t1=d[mode+"_t1"].astype(np.float64)/100; w=d[mode+"_wave"]
    ok=c!=0; cost=(c&0x7FFFFFFF).astype(np.float64); hit=(c>>31)!=0
    span=t1[ok].max(); dur=t1-t0
    print(mode,"span %.0f us; pixels %d; hit flag %.3f"%(span,ok.sum(),hit[ok].mean()))
    print("  start-time percentiles 50/90/99/100 (share of span):",(np.percentile(t0[ok],[50,90,99,100])/span).round(3))
    print("  us per cost unit: pct 10/50/90:",np.percentile(dur[ok]/np.maximum(cost[ok],1),[10,50,90]).round(4)," hit pixels:",np.percentile(dur[ok&hit]/np.maximum(cost[ok&hit],1),[10,50,90]).round(4)," no-hit:",np.percentile(dur[ok&~hit]/np.maximum(cost[ok&~hit],1),[10,50,90]).round(4))
    late=ok&(t1>0.9*span)
    print("  pixels ending in the last 10%% of the span: %d; their start (share of span) pct 10/50/90: %s; duration/span pct 10/50/90: %s; hit flag %.3f; cost pct 10/50/90 %s"%(late.sum(),(np.percentile(t0[late],[10,50,90])/span).round(3),(np.percentile(dur[late],[10,50,90])/span).round(3),hit[late].mean(),np.percentile(cost[late],[10,50,90]).round(0)))
    # duration distribution overall
    print("  duration/span pct 50/90/99/99.9/max:",(np.percentile(dur[ok],[50,90,99,99.9,100])/span).round(3)," cost pct:",np.percentile(cost[ok],[50,90,99,99.9,100]).round(0))
    # the heaviest (by duration) pixels: when did they start?
    heavy=ok&(dur>0.25*span)
    print("  pixels longer than 25%% of the span: %d; start pct 10/50/90/99: %s; hit flag %.3f"%(heavy.sum(),(np.percentile(t0[heavy],[10,50,90,99])/span).round(3),hit[heavy].mean()))
    # correlation between start order and cost
    r=np.argsort(np.argsort(t0[ok])); print("  spearman(start rank, -cost): %.3f"%np.corrcoef(r,-np.argsort(np.argsort(cost[ok])))[0,1])
