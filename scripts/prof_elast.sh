#!/bin/bash
# Elasticity3D 128^3 (config 3) on the block pencil kernel: kernel trace + PMC passes -> gpurun_out/r03e
export TMPDIR=/tmp
OUT=gpurun_out/r03e
rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -o kt -- python3 scripts/bench_configs.py full3 > $OUT/kt.log 2>&1
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d $OUT/pmc_$c -o p -- python3 scripts/bench_configs.py full3 > $OUT/pmc_$c.log 2>&1
done
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F64 GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $OUT/pmc_SQ -o p -- python3 scripts/bench_configs.py full3 > $OUT/pmc_SQ.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/pmc_LDS -o p -- python3 scripts/bench_configs.py full3 > $OUT/pmc_LDS.log 2>&1
python3 scripts/pmc_summary.py $OUT/pmc_FETCH_SIZE $OUT/pmc_WRITE_SIZE $OUT/pmc_SQ $OUT/pmc_LDS > $OUT/pmc_summary.csv
grep -i "block_pencil\|Name" $OUT/kt/kt_kernel_stats.csv | head -5
grep block_pencil $OUT/pmc_summary.csv
