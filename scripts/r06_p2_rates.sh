# Round 6: the p = 2 walks after a change, one lease: config 2, config 4 (fused pair / two calls), CH on the NURBS net
python bench.py --form poisson --degree 2 --size 128 --steps 10 --warmup 3 --no-cpu-baseline --no-live-traffic 2>/dev/null | python -c "
import sys,json; l=json.loads(sys.stdin.read()); r=l['roofline']; print('config 2', round(l['value']/1e6,1), 'M el/s  launch ms', round(r['avg_launch_ms'],4))"
bash scripts/r06_fused.sh
for tc in "" "--two-calls"; do python bench.py --form cahnhilliard --size 128 --geometry $tc --steps 6 --warmup 2 --no-cpu-baseline --no-live-traffic 2>/dev/null | python -c "
import sys,json; l=json.loads(sys.stdin.read()); r=l['roofline']; print('CH 128 NURBS', '$tc' or 'one call', round(l['value']/1e6,1), 'M el/s  launch ms', round(r['avg_launch_ms'],3), l['config']['kernels'][:100])"; done
