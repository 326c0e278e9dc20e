"""Soak of tests/test_gpu_fuzz.py::test_random_discretisation over further seeds (not part of the suite): python scripts/soak_linear.py [first] [last]"""
import os, sys
sys.path.insert(0, "tests"); sys.path.insert(0, "oracle"); sys.path.insert(0, ".")
import numpy as np
import test_gpu_fuzz as T


class _MP:
    def setenv(self, k, v): os.environ[k] = v


first, last = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (150, 450)
bad = 0
for seed in range(first, last):
    for kernel in (0, 1):
        try:
            T.test_random_discretisation(seed, kernel, _MP())
        except Exception as e:
            bad += 1
            print("seed", seed, "kernel", kernel, "FAILED", repr(e)[:300])
print("done, failures:", bad)
