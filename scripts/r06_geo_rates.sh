# Round 6: the Gram walk on the NURBS net at p = 3 (System / Matrix drivers) and p = 2
python bench.py --form poisson --degree 3 --size 128 --geometry --steps 5 --warmup 2 --no-cpu-baseline --no-live-traffic 2>/dev/null | python -c "
import sys,json; l=json.loads(sys.stdin.read()); r=l['roofline']; print('p=3 128^3 NURBS System', round(l['value']/1e6,1), 'M el/s  launch ms', round(r['avg_launch_ms'],3))"
python bench.py --form poisson --degree 3 --size 64 --geometry --steps 5 --warmup 2 --no-cpu-baseline --no-live-traffic 2>/dev/null | python -c "
import sys,json; l=json.loads(sys.stdin.read()); r=l['roofline']; print('p=3 64^3 NURBS System', round(l['value']/1e6,1), 'M el/s  launch ms', round(r['avg_launch_ms'],3))"
python bench.py --form poisson --degree 2 --size 96 --geometry --steps 5 --warmup 2 --no-cpu-baseline --no-live-traffic 2>/dev/null | python -c "
import sys,json; l=json.loads(sys.stdin.read()); r=l['roofline']; print('p=2 96^3 NURBS System', round(l['value']/1e6,1), 'M el/s  launch ms', round(r['avg_launch_ms'],3))"
