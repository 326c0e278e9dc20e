#!/bin/bash
# Collects the evidence behind bench.py's roofline block on the GPU box (run through gpurun from the repo root):
#   1. the bench line itself                      -> gpurun_out/prof/bench_line.json
#   2. rocprofv3 --kernel-trace --stats           -> gpurun_out/prof/kt/...
#   3. rocprofv3 --pmc passes (one counter group per pass, no trace domains besides kernel-trace)
#   4. the same for the Elasticity3D config through scripts/bench_configs.py (feature-GEMM kernel)
# scripts/profile_collect.py then writes the summaries under profiles/.
export TMPDIR=/tmp
OUT=gpurun_out/prof
rm -rf $OUT; mkdir -p $OUT
python3 bench.py --steps 5 --warmup 1 > $OUT/bench_line.json 2> $OUT/bench_line.err
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -o kt -- python3 bench.py --steps 4 --warmup 1 --no-cpu-baseline > $OUT/kt.log 2>&1
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d $OUT/pmc_$c -o p -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline > $OUT/pmc_$c.log 2>&1
done
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F64 GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_SQ -o p -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline > $OUT/pmc_SQ.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/pmc_LDS -o p -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline > $OUT/pmc_LDS.log 2>&1
# secondary: Elasticity3D p=3 at its full config size (feature-GEMM kernel)
BENCH_COMPACT=1 python3 scripts/bench_configs.py c1 c2 full3 full4 full5 c5r c6 c6b > $OUT/configs.txt 2> $OUT/configs.err
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt_elast -o kt -- python3 scripts/bench_configs.py full3 > $OUT/kt_elast.log 2>&1
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d $OUT/pmce_$c -o p -- python3 scripts/bench_configs.py c3 > $OUT/pmce_$c.log 2>&1
done
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F64 GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmce_SQ -o p -- python3 scripts/bench_configs.py c3 > $OUT/pmce_SQ.log 2>&1
# configs 4 and 5 (element mode of the feature kernel): kernel trace + MFMA / HBM counters on a smaller mesh of the same kind
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt_ch -o kt -- python3 scripts/bench_configs.py full4 > $OUT/kt_ch.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt_ns -o kt -- python3 scripts/bench_configs.py full5 > $OUT/kt_ns.log 2>&1
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d $OUT/pmcc_$c -o p -- python3 scripts/bench_configs.py c4 c5 > $OUT/pmcc_$c.log 2>&1
done
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F64 GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmcc_SQ -o p -- python3 scripts/bench_configs.py c4 c5 > $OUT/pmcc_SQ.log 2>&1
BENCH_COMPACT=1 python3 scripts/bench_configs.py c5g c6m c6p c7 >> $OUT/configs.txt 2>> $OUT/configs.err
python3 scripts/bench_rtc.py > $OUT/rtc.txt 2> $OUT/rtc.err
find $OUT -name "*.csv" | head -60
tail -c 600 $OUT/bench_line.json
