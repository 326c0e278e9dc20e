#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc counter_collection CSVs: per (kernel, counter) mean over dispatches.
usage: pmc_summary.py <dir-with-*_counter_collection.csv> [...]

MFMA-busy fraction quoted from these summaries (scripts/profile_collect.py):
    mfma_busy_pmc = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE x 128).
The 128 is a calibration, not a guess: a loop of nothing but v_mfma_f64_16x16x4_f64 that saturates the fp64 matrix pipes
(scripts/mfma_probe.hip, rate<4>: 72.7-74.5 TFLOP/s by HIP events) reports 125.2-126.2 busy cycles per GRBM_GUI_ACTIVE cycle -- the
counter is summed over the device's 1024 pipes (256 CUs x 4 SIMDs) in units of 8 cycles.  profiles/r04_mfma_busy_calibration.txt
holds the run.  SQ_BUSY_CYCLES / GRBM_GUI_ACTIVE sits at 3.9-4.0 whenever a wave is resident and does not measure the pipe."""
import collections
import csv
import glob
import sys

print("Kernel,Counter,Dispatches,MeanPerDispatch")
for d in sys.argv[1:]:
    for f in sorted(glob.glob(d + "/*counter_collection.csv")):
        agg = collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(f)):
            agg[r["Kernel_Name"].split("(")[0]][r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, v in agg.items():
            for c, vals in v.items():
                print('"%s",%s,%d,%.6g' % (k, c, len(vals), sum(vals) / len(vals)))
