# Round 6: config 4 (CahnHilliard 256^3, IFunction + IJacobian) with the Tangent on the pencil walk / the patch walk
for ps in 0 1; do IGX_PATCH_STATE=$ps python bench.py --form cahnhilliard --steps 6 --warmup 2 --no-cpu-baseline --no-live-traffic 2>/dev/null | python -c "
import sys,json; l=json.loads(sys.stdin.read()); r=l['roofline']; print('config 4 IGX_PATCH_STATE=$ps', round(l['value']/1e6,1), 'M el/s  launch ms', round(r['avg_launch_ms'],3), 'launches', r.get('launches'), l['config']['kernels'][:140])"; done
IGX_PATCH_STATE=1 python bench.py --form cahnhilliard --size 128 --steps 6 --warmup 2 --no-cpu-baseline --no-live-traffic 2>/dev/null | python -c "
import sys,json; l=json.loads(sys.stdin.read()); r=l['roofline']; print('CH 128^3 patch', round(l['value']/1e6,1), 'M el/s  launch ms', round(r['avg_launch_ms'],3))"
