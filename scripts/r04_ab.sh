#!/usr/bin/env bash
# Round 4, item 3: the headline kernel of round 2 (worktree ab_r02 = git 80a3c65), of round 3 (ab_libs/libpetiga_amd_r03.so) and of
# this tree, alternating inside ONE gpurun lease (same box, same clocks); shader cycles per launch is the figure to compare.
set -u
mkdir -p gpurun_out
OUT=gpurun_out/r04_headline_ab.txt
: > $OUT
for round in 1 2 3; do
  python scripts/headline_ab.py --pkg ab_r02 --label r02 --steps 8 2>>gpurun_out/ab_err.log | tail -1 >> $OUT
  IGX_LIB=ab_libs/libpetiga_amd_r03.so python scripts/headline_ab.py --label r03 --steps 8 2>>gpurun_out/ab_err.log | tail -1 >> $OUT
  python scripts/headline_ab.py --label r04 --steps 8 2>>gpurun_out/ab_err.log | tail -1 >> $OUT
done
cat $OUT
