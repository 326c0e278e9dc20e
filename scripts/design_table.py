#!/usr/bin/env python3
"""The current-state table of DESIGN.md section 0 from the committed evidence: profiles/<tag>_<workload>_line.json (the bench line of
the profile run) and profiles/traffic.json (the PMC passes of the same gpurun call).  usage: python scripts/design_table.py r05"""
import json
import os
import sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r05"
ROWS = [("poisson", "metric: Poisson p=3 256³ System"), ("poisson_p2", "config 2: Poisson p=2 128³ System"), ("elasticity", "config 3: Elasticity p=3 128³ System"),
        ("cahnhilliard", "config 4: Cahn–Hilliard p=2 256³ IFunction + IJacobian"), ("nsvms", "config 5 (one GPU's share): NS-VMS p=3 96³ on the NURBS net, IFunction + IJacobian"),
        ("poisson_p2_nurbs", "Poisson p=2 96³ on the NURBS net"), ("cahnhilliard_nurbs", "Cahn–Hilliard p=2 128³ on the NURBS net, pair"), ("elasticity_nurbs", "Elasticity p=3 64³ on the NURBS net, System")]
traffic = {c["tag"]: c for c in json.load(open("profiles/traffic.json"))["configs"]}
print("| workload | dominant kernel | M el/s | ms per step | launch ms | executed fraction of 78.6 TF | algorithmic fraction | MFMA busy (PMC) | HBM KB per element | CPU port, el/s (cores) |")
print("|---|---|---|---|---|---|---|---|---|---|")
for w, label in ROWS:
    f = "profiles/%s_%s_line.json" % (tag, w)
    if not os.path.exists(f):
        continue
    l = json.load(open(f)); r = l["roofline"]; t = traffic.get(w, {})
    cb = l.get("cpu_baseline") or {}
    print("| %s | `%s` | **%.1f** | %.1f | %.2f | %.2f | %s | %s | %s | %s |" % (
        label, r.get("traffic_kernel") or r.get("kernel"), l["value"] / 1e6, l["ms_per_step"], r["avg_launch_ms"], r["frac"],
        ("%.2f" % r["frac_algorithmic"]) if r.get("frac_algorithmic") is not None else "—",
        ("%.2f" % t["mfma_busy_pmc"]) if t.get("kernel_tag") == tag and t.get("mfma_busy_pmc") is not None else "—",
        ("%.1f" % (t["bytes_per_element"] / 1e3)) if t.get("kernel_tag") == tag else "—",
        ("%.0f (%d)" % (cb["value"], cb["cores"])) if cb else "—"))
