#!/usr/bin/env bash
# N ranks as processes sharing one GPU (gloo test transport): bench.py's whole multi-rank flow -- partition, refresh, assembly, ghost-row
# reduction, checksum against the single-rank assembly -- for rank counts and workloads beyond what the test suite runs.
run() { IGX_BENCH_BACKEND=gloo timeout 900 python bench.py "$@" --steps 1 --warmup 1 --no-cpu-baseline 2>gpurun_out/mr.err | python -c "
import json,sys
l=sys.stdin.readline()
try:
    d=json.loads(l); c=d['config']; print('OK', '$*', c['partition'], c['kernels'][:40], 'early', c['exchange_early_phases'], 'maxdiff %.1e' % max(c['checksum_check']['rel_diff']))
except Exception as e:
    print('FAIL', '$*', repr(e)); print(open('gpurun_out/mr.err').read()[-1500:])"; }
if [ "${THIN:-0}" != 1 ]; then
for n in 3 5 6 7; do run --gpus $n --form poisson --size 48; done
run --gpus 6 --form cahnhilliard --size 48
run --gpus 3 --form nsvms --size 24
run --gpus 5 --form elasticity --size 40
run --gpus 8 --form elasticity --size 32
run --gpus 8 --form poisson --size 16
run --gpus 2 --form poisson --size 256
fi
# ranks thinner than p elements on the split axis: ghost rows travel two ranks up
if [ "${THIN:-0}" = 1 ]; then
run --gpus 7 --form poisson --size 14
run --gpus 5 --form cahnhilliard --size 10
run --gpus 6 --form elasticity --size 12
fi
