# Round 6: the Tangent walk (CH p = 2) at small sizes: the launcher's segments against forced ones (pair M el/s | Tangent launch ms)
for sz in 32 48 64; do
for n in 0 2 3 4 6 8 12 16; do echo -n "size $sz IGX_NSEG=$n: "; IGX_NSEG=$n python bench.py --form cahnhilliard --size $sz --steps 6 --warmup 2 --no-cpu-baseline --no-live-traffic 2>/dev/null | python -c "
import sys,json; l=json.loads(sys.stdin.read()); r=l['roofline']; print(round(l['value']/1e6,1), '|', round(r['avg_launch_ms'],4))"; done; done
