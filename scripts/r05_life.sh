# -DIGX_DEBUG build: the life of a workgroup of the pencil walk (staging | lane set-up | walk | trailing leaves), first launch of each workload
export IGX_USE_DEBUG_LIB=1 IGX_DEBUG_TIMING=1
for a in "--form poisson --degree 2 --size 128" "--form poisson" "--form cahnhilliard --size 128" "--form poisson --size 64 --geometry" "--form poisson --size 128 --geometry"; do
  echo "== $a"; python bench.py $a --steps 2 --warmup 1 --no-cpu-baseline --no-live-traffic 2>&1 >/dev/null | grep "igx pencil timing" | grep -v histogram | head -12
done
