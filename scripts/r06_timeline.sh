# Round 6: where a config-2 launch spends the time outside its workgroups' lives (-DIGX_DEBUG build: per-CU timeline on the 100 MHz clock).
# IGX_DEBUG_TIMING=n stamps the n-th pencil launch of the process: 1 = the first launch of the first assembly, 12 = a steady-state one.
export IGX_USE_DEBUG_LIB=1
for n in 1 12 14; do
  echo "== config 2 (poisson p=2 128^3), launch $n"
  IGX_DEBUG_TIMING=$n python bench.py --form poisson --degree 2 --size 128 --steps 3 --warmup 2 --no-cpu-baseline --no-live-traffic 2>&1 >/dev/null | grep "igx pencil timing" | grep -v histogram
done
echo "== config 2 release build: launch ms and shader clock"
unset IGX_USE_DEBUG_LIB
IGX_CLOCK_PROBE=1 python bench.py --form poisson --degree 2 --size 128 --steps 10 --warmup 3 --no-cpu-baseline --no-live-traffic 2>/dev/null | python -c "
import sys,json; l=json.loads(sys.stdin.read()); r=l['roofline']; print('M el/s', round(l['value']/1e6,1), 'ms/step', round(l['ms_per_step'],3), 'launch ms', round(r['avg_launch_ms'],4), 'launches', r.get('launches'), 'shader MHz', r.get('shader_clock_mhz'))"
echo "== config 4 (cahnhilliard 256^3), launch 12"
IGX_USE_DEBUG_LIB=1 IGX_DEBUG_TIMING=12 python bench.py --form cahnhilliard --steps 2 --warmup 1 --no-cpu-baseline --no-live-traffic 2>&1 >/dev/null | grep "igx pencil timing" | grep -v histogram
