# segment-count sweep of the packed p = 2 kernels (IGX_NSEG: experiment switch of launch_pencils)
for n in 0 1 2 3 4 5 6 8; do IGX_NSEG=$n python bench.py --form poisson --degree 2 --size 128 --steps 10 --warmup 2 --no-cpu-baseline --no-live-traffic 2>/dev/null | python -c "
import sys,json; l=json.loads(sys.stdin.read()); r=l['roofline']; print('poisson p2 128 nseg', $n, round(l['value']/1e6,1), 'M el/s  launch ms', round(r['avg_launch_ms'],3))"; done
for n in 0 1 2 3 4; do IGX_NSEG=$n python bench.py --form cahnhilliard --size 128 --steps 6 --warmup 2 --no-cpu-baseline --no-live-traffic 2>/dev/null | python -c "
import sys,json; l=json.loads(sys.stdin.read()); r=l['roofline']; print('CH 128 nseg', $n, round(l['value']/1e6,1), 'M el/s  launch ms', round(r['avg_launch_ms'],3))"; done
for n in 0 2 3 4 6; do IGX_NSEG=$n python bench.py --form cahnhilliard --steps 4 --warmup 1 --no-cpu-baseline --no-live-traffic 2>/dev/null | python -c "
import sys,json; l=json.loads(sys.stdin.read()); r=l['roofline']; print('CH 256 nseg', $n, round(l['value']/1e6,1), 'M el/s  launch ms', round(r['avg_launch_ms'],3))"; done
