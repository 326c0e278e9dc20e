# Round 6: the patch walk (IGX_PATCH=1) against the pencil walk, Poisson p=2 Matrix driver (no F, no Dirichlet fix-up), sizes 64..256
python - <<'PY'
import os, sys, time
sys.path.insert(0, ".")
for patch in ("0", "1"):
    os.environ["IGX_PATCH"] = patch
    import importlib, petiga_amd as P
    for n in (64, 128, 256):
        g = P.IGX(3, 1)
        for i in range(3): g.axis_uniform(i, 2, n)
        g.setup(); g.set_form("poisson")
        A = g.create_mat()
        g.set_timing(True)
        for d in range(3):
            for sd in range(2): g.set_boundary_value(d, sd, 0, 1.0)
        b = g.create_vec()
        for drv in ("Matrix", "System"):
            run = (lambda: g.compute_matrix(A)) if drv == "Matrix" else (lambda: g.compute_system(A, b))
            for _ in range(3): run()
            g.synchronize()
            t = []
            for _ in range(6):
                t0 = time.perf_counter(); run(); g.synchronize(); t.append(time.perf_counter() - t0)
            print("IGX_PATCH=%s %d^3 %s: %.3f ms (min of 6) = %.1f M el/s  %s" % (patch, n, drv, min(t) * 1e3, n ** 3 / min(t) / 1e6, g.kernel_name()[:60]), flush=True)
        del A, b, g
PY
