# Round 6: where the state patch walk's step goes: timing-only switches (wrong results) on config 4 at 128^3
for d in 0 1 2 4 8 3 7 15 9; do IGX_PATCH_DBG=$d IGX_PATCH_STATE=1 python bench.py --form cahnhilliard --size 128 --steps 4 --warmup 1 --no-cpu-baseline --no-live-traffic 2>/dev/null | python -c "
import sys,json; l=json.loads(sys.stdin.read()); r=l['roofline']; print('dbg $d launch ms', round(r['avg_launch_ms'],3))"; done
