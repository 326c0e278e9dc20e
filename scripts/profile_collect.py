#!/usr/bin/env python3
"""Turn the output of scripts/profile_round.sh (gpurun_out/prof) into the committed evidence under profiles/.
usage: python scripts/profile_collect.py <round-tag, e.g. r01>"""
import collections
import csv
import glob
import json
import os
import shutil
import sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r02"
SRC, DST = "gpurun_out/prof", "profiles"


def pmc(dirs):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for d in dirs:
        for f in sorted(glob.glob(d + "/*counter_collection.csv")):
            for r in csv.DictReader(open(f)):
                agg[r["Kernel_Name"].split("(")[0]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return agg


def write_pmc(path, agg):
    with open(path, "w") as f:
        f.write("Kernel,Counter,Dispatches,MeanPerDispatch\n")
        for k, v in sorted(agg.items()):
            for c, vals in sorted(v.items()):
                f.write('"%s",%s,%d,%.6g\n' % (k, c, len(vals), sum(vals) / len(vals)))


shutil.copy(SRC + "/kt/kt_kernel_stats.csv", "%s/%s_bench256_kernel_stats.csv" % (DST, tag))
shutil.copy(SRC + "/kt_elast/kt_kernel_stats.csv", "%s/%s_elasticity128_kernel_stats.csv" % (DST, tag))
shutil.copy(SRC + "/configs.txt", "%s/%s_secondary_configs.txt" % (DST, tag))
for src, dst in (("kt_ch", "cahnhilliard256"), ("kt_ns", "nsvms96")):
    if os.path.exists(SRC + "/%s/kt_kernel_stats.csv" % src):
        shutil.copy(SRC + "/%s/kt_kernel_stats.csv" % src, "%s/%s_%s_kernel_stats.csv" % (DST, tag, dst))
if glob.glob(SRC + "/pmcc_SQ/*counter_collection.csv"):
    write_pmc("%s/%s_ch128_nsvms32_pmc_summary.csv" % (DST, tag), pmc([SRC + "/pmcc_" + c for c in ("FETCH_SIZE", "WRITE_SIZE", "SQ")]))
if os.path.exists(SRC + "/rtc.txt"):
    shutil.copy(SRC + "/rtc.txt", "%s/%s_runtime_forms.txt" % (DST, tag))
line = json.loads(open(SRC + "/bench_line.json").read().strip().splitlines()[-1])
a = pmc([SRC + "/pmc_" + c for c in ("FETCH_SIZE", "WRITE_SIZE", "SQ", "LDS")])
write_pmc("%s/%s_bench256_pmc_summary.csv" % (DST, tag), a)
write_pmc("%s/%s_elasticity64_pmc_summary.csv" % (DST, tag), pmc([SRC + "/pmce_" + c for c in ("FETCH_SIZE", "WRITE_SIZE", "SQ")]))
dom = [k for k in a if "gram_pencil" in k]
if dom:
    k = dom[0]
    mean = lambda c: sum(a[k][c]) / len(a[k][c])
    F, W = mean("FETCH_SIZE"), mean("WRITE_SIZE")
    r = line["roofline"]
    traffic = dict(round=int(tag[1:]), kernel_tag=tag, size=256, degree=3, n_gpus=1, kernel=k.replace("void igx::", ""), launches_per_step=r["launches_per_step"],
                   FETCH_SIZE_KB_per_launch=F, WRITE_SIZE_KB_per_launch=W, raw_bytes_per_launch=(F + W) * 1024, bytes_per_launch=(2 * F + W) * 1024,
                   mfma_busy_pmc=mean("SQ_VALU_MFMA_BUSY_CYCLES") / mean("GRBM_GUI_ACTIVE") / 128.0,
                   note="rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes (profiles/%s_bench256_pmc_summary.csv); "
                        "hbm_bytes=(FETCH_SIZE+WRITE_SIZE)*1024 with FETCH_SIZE doubled (gfx950 reports half of a wide coalesced read, "
                        "MI355X_MICROARCH.md HBM section); algorithmic read-modify-write bytes of the launch: 1 048 576 elements x 1792 entries "
                        "x 8 B x 2 = 30.1e9; compulsory (write-once) bytes 2785 B/element = 2.9e9; mfma_busy_pmc = SQ_VALU_MFMA_BUSY_CYCLES / "
                        "(GRBM_GUI_ACTIVE * 128)" % tag)
    json.dump(traffic, open(DST + "/traffic.json", "w"), indent=1)
    line["roofline"]["traffic"] = traffic["bytes_per_launch"]
    line["roofline"]["traffic_source"] = "rocprofv3 --pmc passes of the same command in the same gpurun call (profiles/%s_bench256_pmc_summary.csv)" % tag
    if r.get("avg_launch_ms"):
        line["roofline"]["hbm_frac"] = traffic["bytes_per_launch"] / (r["avg_launch_ms"] * 1e-3) / 8e12
json.dump(line, open("%s/%s_bench256_line.json" % (DST, tag), "w"), indent=1)
print(open(DST + "/traffic.json").read())
