# Round 6: the vector-only drivers after the pipelined walk (vec_sumfact): config 4's IFunction, CH on the NURBS net, Elasticity with a body force
python bench.py --form cahnhilliard --steps 6 --warmup 2 --no-cpu-baseline --no-live-traffic 2>/dev/null | python -c "
import sys,json; l=json.loads(sys.stdin.read()); r=l['roofline']; print('config 4 pair', round(l['value']/1e6,1), 'M el/s  ms/step', round(l['ms_per_step'],2), 'Tangent launch ms', round(r['avg_launch_ms'],3))"
python bench.py --form cahnhilliard --size 128 --geometry --steps 6 --warmup 2 --no-cpu-baseline --no-live-traffic 2>/dev/null | python -c "
import sys,json; l=json.loads(sys.stdin.read()); print('CH 128 NURBS pair', round(l['value']/1e6,1), 'M el/s  ms/step', round(l['ms_per_step'],2))"
BENCH_COMPACT=1 python scripts/bench_configs.py full4 2>/dev/null | tail -3
