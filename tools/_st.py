import time
from tdt4230_project_raytracing_amd import rt
ctx = rt.Context()
for m in (9, 10, 11, 12):
    t = time.time(); v = ctx.selftest(m); print("selftest", m, v, "%.1fs" % (time.time() - t), flush=True)
