#!/usr/bin/env python3
"""Turn the output of scripts/profile_round.sh (gpurun_out/prof) into the committed evidence under profiles/.
usage: python scripts/profile_collect.py <round-tag, e.g. r03> [workload tags ...]
(with workload tags only those are collected -- gpurun_out/prof merges over what earlier calls left there, so a subset run must
name its subset or stale files of other workloads would be re-labelled with this round's tag)

Per workload: <tag>_<form>_line.json (the bench line; its roofline.traffic filled in from the PMC passes of the same gpurun
call), <tag>_<form>_kernel_stats.csv (rocprofv3 --kernel-trace --stats), <tag>_<form>_pmc_summary.csv (mean per dispatch of every
counter); profiles/traffic.json is what bench.py replays as `roofline.traffic` in later runs."""
import collections
import csv
import glob
import json
import os
import shutil
import sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r05"
ONLY = set(sys.argv[2:])
SRC, DST = "gpurun_out/prof", "profiles"
SIZES = {"poisson": 256, "poisson_p2": 128, "poisson_p2_nurbs": 96, "elasticity": 128, "cahnhilliard": 256, "nsvms": 96, "cahnhilliard_nurbs": 128, "elasticity_nurbs": 64}
KEY = {"poisson": "gram_pencil", "poisson_p2": "gram_p", "poisson_p2_nurbs": "gram_pencil", "elasticity": "block_pencil", "cahnhilliard": "state_pencil", "nsvms": "band_pt<", "cahnhilliard_nurbs": "state_pencil", "elasticity_nurbs": "band_pt<"}
FORM = {"poisson_p2": "poisson", "poisson_p2_nurbs": "poisson", "cahnhilliard_nurbs": "cahnhilliard", "elasticity_nurbs": "elasticity"}


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


configs = []
for form in ("poisson", "poisson_p2", "poisson_p2_nurbs", "elasticity", "elasticity_nurbs", "cahnhilliard", "nsvms", "cahnhilliard_nurbs"):
    lf = "%s/line_%s.json" % (SRC, form)
    if ONLY and form not in ONLY:
        continue
    if not os.path.exists(lf) or not open(lf).read().strip():
        continue
    line = json.loads(open(lf).read().strip().splitlines()[-1])
    if os.path.exists("%s/kt_%s/kt_kernel_stats.csv" % (SRC, form)):
        shutil.copy("%s/kt_%s/kt_kernel_stats.csv" % (SRC, form), "%s/%s_%s_kernel_stats.csv" % (DST, tag, form))
    a = pmc(["%s/pmc_%s_%s" % (SRC, form, c) for c in ("FETCH_SIZE", "WRITE_SIZE", "SQ", "LDS")])
    if a:
        write_pmc("%s/%s_%s_pmc_summary.csv" % (DST, tag, form), a)
    r = line["roofline"]
    # the dominant kernel of the bench line: the IJacobian's for the two-assembly steps (the residual kernel is the vector-only
    # instantiation: HASM = false, "Lb0" in the 7th template argument -- pick the one with the most MFMA work)
    dom = [k for k in a if KEY[form] in k and "SQ_VALU_MFMA_BUSY_CYCLES" in a[k] and "FETCH_SIZE" in a[k]]
    dom.sort(key=lambda k: -sum(a[k]["SQ_VALU_MFMA_BUSY_CYCLES"]) / len(a[k]["SQ_VALU_MFMA_BUSY_CYCLES"]))
    if dom:
        k = dom[0]
        mean = lambda c: sum(a[k][c]) / len(a[k][c])
        F, W = mean("FETCH_SIZE"), mean("WRITE_SIZE")
        ent = dict(form=FORM.get(form, form), tag=form, geometry=form.endswith("_nurbs") or form == "nsvms", round=int(tag[1:]), kernel_tag=tag, size=SIZES[form], n_gpus=1, kernel=k.replace("void igx::", ""),
                   launches_per_step=r["launches_per_step"], FETCH_SIZE_KB_per_launch=F, WRITE_SIZE_KB_per_launch=W,
                   raw_bytes_per_launch=(F + W) * 1024, bytes_per_launch=(2 * F + W) * 1024,
                   bytes_per_element=(2 * F + W) * 1024 / max(r["elements_per_launch"], 1),
                   mfma_busy_pmc=mean("SQ_VALU_MFMA_BUSY_CYCLES") / mean("GRBM_GUI_ACTIVE") / 128.0,
                   note="rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes (profiles/%s_%s_pmc_summary.csv); bytes = (2 x FETCH_SIZE + "
                        "WRITE_SIZE) x 1024: gfx950 reports half of a wide coalesced read (MI355X_MICROARCH.md, HBM section); mfma_busy_pmc = "
                        "SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE x 128)" % (tag, form))
        if form.startswith("poisson"):
            ent["degree"] = 2 if "_p2" in form else 3
        configs.append(ent)
        # (the line keeps the traffic bench.py measured inside its own run; the separate PMC passes of this call go into traffic.json and
        #  next to it as a cross-check)
        line["roofline"]["traffic_pmc_passes"] = ent["bytes_per_launch"]
        line["roofline"]["mfma_busy_pmc"] = ent["mfma_busy_pmc"]
        if line["roofline"].get("traffic") is None:
            line["roofline"]["traffic"] = ent["bytes_per_launch"]
            line["roofline"]["traffic_source"] = "rocprofv3 --pmc passes of the same command in the same gpurun call (profiles/%s_%s_pmc_summary.csv)" % (tag, form)
            if r.get("avg_launch_ms"):
                line["roofline"]["hbm_frac"] = ent["bytes_per_launch"] / (r["avg_launch_ms"] * 1e-3) / 8e12
    json.dump(line, open("%s/%s_%s_line.json" % (DST, tag, form), "w"), indent=1)
if not ONLY and os.path.exists(SRC + "/line_poisson_source.json") and open(SRC + "/line_poisson_source.json").read().strip():
    shutil.copy(SRC + "/line_poisson_source.json", "%s/%s_poisson_source_line.json" % (DST, tag))
for src, dst in (("configs.txt", "secondary_configs.txt"), ("rtc.txt", "runtime_forms.txt")):
    if not ONLY and os.path.exists(SRC + "/" + src):
        shutil.copy(SRC + "/" + src, "%s/%s_%s" % (DST, tag, dst))
if configs:
    # (a run over a subset of the workloads -- TAGS=... scripts/profile_round.sh -- keeps the other entries of the same round)
    try:
        old = json.load(open(DST + "/traffic.json"))
        have = set(c["tag"] for c in configs)      # (entries carry their own kernel_tag: an older round's figure stays until its workload is re-measured)
        configs = [c for c in old.get("configs", []) if c.get("tag") not in have] + configs
    except (OSError, ValueError):
        pass
    json.dump(dict(round=int(tag[1:]), kernel_tag=tag, configs=configs), open(DST + "/traffic.json", "w"), indent=1)
    print(open(DST + "/traffic.json").read())
