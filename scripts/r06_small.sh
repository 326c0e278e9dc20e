# Round 6: small meshes: wall per assembly against the kernels' own time and the HIP calls of the host (one lease)
export TMPDIR=/tmp
mkdir -p gpurun_out/small
python3 scripts/r06_small.py > gpurun_out/small/wall.txt 2>&1
REPS=10 rocprofv3 --kernel-trace --hip-trace --stats --output-format csv -d gpurun_out/small/kt -o kt -- python3 scripts/r06_small.py > gpurun_out/small/kt.log 2>&1
cat gpurun_out/small/wall.txt
python3 - <<'PY'
import csv, glob
for pat in ("*kernel_stats.csv", "*hip_api_stats.csv", "*hip_stats.csv"):
    for f in glob.glob("gpurun_out/small/kt/**/" + pat, recursive=True):
        print("==", f)
        rows = list(csv.DictReader(open(f)))
        for r in rows[:12]:
            print(r["Name"][:70], r["Calls"], r["TotalDurationNs"], r["AverageNs"])
PY
ls -R gpurun_out/small/kt | head -20
