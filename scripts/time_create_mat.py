import time, sys
sys.path.insert(0, ".")
import petiga_amd as P
g = P.IGX(3, 1)
for i in range(3): g.axis_uniform(i, 3, 256)
g.setup()
for k in range(2):
    g.synchronize(); t = time.perf_counter(); A = g.create_mat(); g.synchronize(); print("IGXCreateMat %.1f ms" % ((time.perf_counter() - t) * 1e3)); del A
