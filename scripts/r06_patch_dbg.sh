# Round 6: what the System driver costs the patch walk (IGX_PATCH_DBG bits switch its parts off; wrong results, timing only)
for d in 0 1 2 3 4 8 15; do echo -n "IGX_PATCH_DBG=$d: "; IGX_PATCH=1 IGX_PATCH_DBG=$d python - <<'PY'
import os, sys, time
sys.path.insert(0, ".")
import petiga_amd as P
n = 128
g = P.IGX(3, 1)
for i in range(3): g.axis_uniform(i, 2, n)
g.setup(); g.set_form("poisson")
for d in range(3):
    for sd in range(2): g.set_boundary_value(d, sd, 0, 1.0)
A, b = g.create_mat(), g.create_vec()
for _ in range(3): g.compute_system(A, b)
g.synchronize()
t = []
for _ in range(6):
    t0 = time.perf_counter(); g.compute_system(A, b); g.synchronize(); t.append(time.perf_counter() - t0)
print("128^3 System: %.3f ms" % (min(t) * 1e3))
PY
done
