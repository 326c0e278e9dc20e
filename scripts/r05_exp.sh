export IGX_USE_DEBUG_LIB=1
echo "== headline: shader clock with / without the read-add-write stream (debug build)"
for nf in 0 1; do IGX_DEBUG_NOFLUSH=$nf python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-live-traffic 2>/dev/null | python -c "
import sys,json; l=json.loads(sys.stdin.read()); r=l['roofline']; print('noflush', $nf, 'ms/step', round(l['ms_per_step'],2), 'launch ms', round(r['avg_launch_ms'],3), 'shader MHz', r.get('shader_clock_mhz'), 'per-step MHz', l['per_step'].get('shader_mhz'))"; done
echo "== stamps"
IGX_DEBUG_TIMING=1 python bench.py --form poisson --degree 2 --size 128 --steps 2 --warmup 1 --no-cpu-baseline --no-live-traffic 2>&1 >/dev/null | grep "igx pencil" | head -3
IGX_DEBUG_TIMING=1 python bench.py --form cahnhilliard --size 128 --steps 2 --warmup 1 --no-cpu-baseline --no-live-traffic 2>&1 >/dev/null | grep "igx pencil" | head -3
IGX_DEBUG_TIMING=1 python bench.py --form cahnhilliard --size 128 --geometry --steps 2 --warmup 1 --no-cpu-baseline --no-live-traffic 2>&1 >/dev/null | grep "igx pencil" | head -3
IGX_P2_PACK=0 IGX_DEBUG_TIMING=1 python bench.py --form cahnhilliard --size 128 --steps 2 --warmup 1 --no-cpu-baseline --no-live-traffic 2>&1 >/dev/null | grep "igx pencil" | head -3
