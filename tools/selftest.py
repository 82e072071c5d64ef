import sys; sys.path.insert(0,'/root/repo')
from tdt4230_project_raytracing_amd import rt
import time
with rt.Context(0) as c:
    for w in (0,1,2,3,4,5,6):
        t=time.time(); print(w, c.selftest(w), f"{time.time()-t:.2f}s")
