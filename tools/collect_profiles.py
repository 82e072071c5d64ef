"""copy what tools/profile_r04.sh (ROUND=r03 / r02: the earlier rounds' scripts) left under gpurun_out/<tag>/ into profiles/ under the round's names (r04_*) and refresh
profiles/traffic.json; usage: python tools/collect_profiles.py <tag> [--bench-only]"""
import glob, json, os, shutil, sys
tag = sys.argv[1]
RND = os.environ.get("ROUND", "r04")
PREV = {"r04": "_r03", "r03": "_r02"}.get(RND, "_prev")
bench_only = "--bench-only" in sys.argv
src = os.path.join("gpurun_out", tag)
dst = "profiles"
for s, d in (("bench_c2", RND + "_bench_config2"), ("bench_c3", RND + "_bench_config3"), ("bench_c5", RND + "_bench_config5"),
             ("bench_c5_progressive", RND + "_bench_config5_progressive16"), ("bench_c4_8k", RND + "_bench_config4_8k")):
    p = os.path.join(src, s + ".json")
    if os.path.exists(p):
        line = open(p).read().strip().splitlines()[-1]
        json.loads(line)
        open(os.path.join(dst, d + ".json"), "w").write(line + "\n")
if bench_only:
    sys.exit(0)
for c in (2, 3, 5, 0):
    st = sorted(glob.glob(os.path.join(src, "stats_c%d" % c, "*", "*kernel_stats.csv")))
    if st:
        shutil.copy(st[-1], os.path.join(dst, RND + "_config%d_kernel_stats.csv" % c))
    for mode, suffix in (("fresh", ""), ("replay", "_replay")):
        p = os.path.join(src, "pmc_c%d_%s.summary.txt" % (c, mode))
        if os.path.exists(p):
            json.load(open(p))
            shutil.copy(p, os.path.join(dst, RND + "_config%d_pmc_sq%s.json" % (c, suffix)))
    p = os.path.join(src, "tcc_c%d.summary.txt" % c)
    if os.path.exists(p):
        json.load(open(p))
        shutil.copy(p, os.path.join(dst, RND + "_config%d_pmc_tcc.json" % c))
shutil.copy(os.path.join(src, "pmc_summary.json"), os.path.join(dst, RND + "_pmc_summary.json"))
for p in glob.glob(os.path.join(src, "stats_c*_*.json")):      # pass statistics of the product kernels (tools/loss_budget.py collect)
    if os.path.getsize(p):
        shutil.copy(p, os.path.join(dst, RND + "_" + os.path.basename(p)))
for s_, d_ in (("demo_time.txt", RND + "_demo_time.txt"),):
    if os.path.exists(os.path.join(src, s_)):
        shutil.copy(os.path.join(src, s_), os.path.join(dst, d_))
tr = json.load(open(os.path.join(dst, "traffic.json")))
for c in (2, 3, 5, 0):
    t = json.load(open(os.path.join(dst, RND + "_config%d_pmc_tcc.json" % c)))["counters"]
    for k in ("config%d_spp64_gpus1" % c, "config%d_spp4_gpus1" % c):      # keep the previous round's figures beside the new ones
        if k in tr and (k + PREV) not in tr:
            tr[k + PREV] = tr[k]
    tr["config%d_spp%d_gpus1" % (c, 4 if c == 0 else 64)] = {"FETCH_SIZE_KB": t["FETCH_SIZE"], "WRITE_SIZE_KB": t["WRITE_SIZE"],
                                       "hbm_bytes_per_launch": int((2 * t["FETCH_SIZE"] + t["WRITE_SIZE"]) * 1024),
                                       "TCC_HIT_sum": t["TCC_HIT_sum"], "TCC_MISS_sum": t["TCC_MISS_sum"],
                                       "TCP_TCC_READ_REQ_sum": t.get("TCP_TCC_READ_REQ_sum"), "TCC_EA0_RDREQ_sum": t.get("TCC_EA0_RDREQ_sum")}
json.dump(tr, open(os.path.join(dst, "traffic.json"), "w"), indent=1)
print("profiles refreshed from", src)
