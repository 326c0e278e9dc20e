#!/bin/bash
# Collects the evidence behind bench.py's lines on the GPU box (run through gpurun from the repo root), rounds 4-6:
#   for each workload tag (poisson = the metric; poisson_p2 = config 2; poisson_p2_nurbs = config 2 on the bench's NURBS map;
#   elasticity, cahnhilliard, nsvms = configs 3, 4, 5; cahnhilliard_nurbs = config 4's forms at 128^3 on the bench's NURBS map; TAGS="..."
#   runs a subset and skips the secondary timings):
#     1. the bench line itself (roofline + cpu_baseline, traffic measured in the run)   -> gpurun_out/prof/line_<tag>.json
#     2. rocprofv3 --kernel-trace --stats of the same command                            -> gpurun_out/prof/kt_<tag>/
#     3. rocprofv3 --pmc passes, one counter group per pass (no trace domains besides kernel-trace): FETCH_SIZE, WRITE_SIZE, SQ
# scripts/profile_collect.py then writes the summaries under profiles/.
export TMPDIR=/tmp
OUT=gpurun_out/prof
rm -rf $OUT; mkdir -p $OUT
[ -n "$TAGS" ] && ONLY_TAGS=1
TAGS="${TAGS:-poisson poisson_p2 poisson_p2_nurbs elasticity elasticity_nurbs cahnhilliard nsvms cahnhilliard_nurbs}"
args_of() {
  case $1 in
    poisson) echo "--form poisson" ;;
    poisson_p2) echo "--form poisson --degree 2 --size 128" ;;
    poisson_p2_nurbs) echo "--form poisson --degree 2 --size 96 --geometry" ;;
    cahnhilliard_nurbs) echo "--form cahnhilliard --size 128 --geometry" ;;
    elasticity_nurbs) echo "--form elasticity --size 64 --geometry" ;;
    *) echo "--form $1" ;;
  esac
}
for t in $TAGS; do
  A=$(args_of $t)
  python3 bench.py $A --steps 10 --warmup 2 > $OUT/line_$t.json 2> $OUT/line_$t.err
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt_$t -o kt -- python3 bench.py $A --steps 3 --warmup 1 --no-cpu-baseline --no-live-traffic > $OUT/kt_$t.log 2>&1
  for c in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --kernel-trace --pmc $c --output-format csv -d $OUT/pmc_${t}_$c -o p -- python3 bench.py $A --steps 1 --warmup 0 --no-cpu-baseline --no-live-traffic > $OUT/pmc_${t}_$c.log 2>&1
  done
  rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F64 GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_${t}_SQ -o p -- python3 bench.py $A --steps 1 --warmup 0 --no-cpu-baseline --no-live-traffic > $OUT/pmc_${t}_SQ.log 2>&1
done
if [ -n "$ONLY_TAGS" ]; then find $OUT -name "*.csv" | wc -l; for t in $TAGS; do tail -c 300 $OUT/line_$t.json; echo; done; exit 0; fi
rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/pmc_poisson_LDS -o p -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-live-traffic > $OUT/pmc_poisson_LDS.log 2>&1
# the metric configuration's form given as run-time source
python3 bench.py --source --steps 10 --warmup 2 --no-cpu-baseline > $OUT/line_poisson_source.json 2> $OUT/line_poisson_source.err
# secondary timings (one assembly each, current kernels)
BENCH_COMPACT=1 python3 scripts/bench_configs.py c1 c2 full3 full4 c4g full5 c5r c5g c6 c6b c6m c6p c7 > $OUT/configs.txt 2> $OUT/configs.err
python3 scripts/bench_rtc.py > $OUT/rtc.txt 2> $OUT/rtc.err
find $OUT -name "*.csv" | wc -l
for t in $TAGS; do tail -c 300 $OUT/line_$t.json; echo; done
