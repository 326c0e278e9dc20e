# Round 6: Elasticity off the identity geometry and with a body force (band_pt with Gram pairs; block_pencil + a vector pass)
run() { echo -n "$1: "; python bench.py $1 --steps 3 --warmup 1 --no-cpu-baseline --no-live-traffic 2>/dev/null | python -c "
import sys,json; l=json.loads(sys.stdin.read()); r=l['roofline']; print(round(l['value']/1e6,2), 'M el/s  ms/step', round(l['ms_per_step'],2), ' launch ms', round(r['avg_launch_ms'],3), 'frac', round(r['frac'],3), l['config']['kernels'][:70])"; }
run "--form elasticity --size 128"
run "--form elasticity --size 128 --body-force"
run "--form elasticity --size 64 --geometry"
run "--form elasticity --size 64 --geometry --body-force"
run "--form elasticity --size 96 --geometry"
run "--form elasticity --size 128 --geometry"
run "--form elasticity --size 64 --geometry --kernel 3"
