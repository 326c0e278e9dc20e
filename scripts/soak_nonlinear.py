import sys, os
sys.path.insert(0, "tests"); sys.path.insert(0, "oracle"); sys.path.insert(0, ".")
import numpy as np
import test_gpu_fuzz as T
bad = 0
for seed in range(60, 400):
    try:
        T.test_random_nonlinear_discretisation(seed)
    except Exception as e:
        bad += 1
        print("seed", seed, "FAILED", repr(e)[:300])
print("done, failures:", bad)
