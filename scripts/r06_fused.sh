# Round 6: the fused IFunction + IJacobian pass (state_pencil_kr) against the two drivers, config 4 (256^3) and 128^3
for sz in 128 256; do for tc in "" "--two-calls"; do
  python bench.py --form cahnhilliard --size $sz $tc --steps 6 --warmup 2 --no-cpu-baseline --no-live-traffic 2>/dev/null | python -c "
import sys,json; l=json.loads(sys.stdin.read()); r=l['roofline']; print('CH $sz', '$tc' or 'fused', round(l['value']/1e6,1), 'M el/s  ms/step', round(l['ms_per_step'],2), ' launch ms', round(r['avg_launch_ms'],3), l['config']['kernels'][:90])"
done; done
