#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc counter_collection CSVs: per (kernel, counter) mean over dispatches.
usage: pmc_summary.py <dir-with-*_counter_collection.csv> [...]"""
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
