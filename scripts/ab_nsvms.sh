for i in 1 2; do
for lib in "" ab_libs/libpetiga_amd_nopq.so; do
for n in 24 8; do
  IGX_LIB=$lib IGX_NSEG=$n python bench.py --form nsvms --steps 4 --warmup 1 --no-cpu-baseline --no-live-traffic 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('lib=$lib nseg=$n ms/step %.2f' % d['ms_per_step'])"
done; done; done
