# Round 6: phase stamps of the Gram walk on the NURBS net (p = 2 96^3, p = 3 128^3; -DIGX_DEBUG build, a steady-state launch)
export IGX_USE_DEBUG_LIB=1 IGX_DEBUG_TIMING=12
for nf in 0 2; do
  echo "== p=2 96^3 NURBS IGX_DEBUG_NOFLUSH=$nf"
  IGX_DEBUG_NOFLUSH=$nf python bench.py --form poisson --degree 2 --size 96 --geometry --steps 3 --warmup 2 --no-cpu-baseline --no-live-traffic 2>&1 >/dev/null | grep "igx pencil timing" | grep -v histogram | head -1
done
export IGX_DEBUG_TIMING=20
for nf in 0 2; do
  echo "== p=3 128^3 NURBS IGX_DEBUG_NOFLUSH=$nf"
  IGX_DEBUG_NOFLUSH=$nf python bench.py --form poisson --degree 3 --size 128 --geometry --steps 3 --warmup 2 --no-cpu-baseline --no-live-traffic 2>&1 >/dev/null | grep "igx pencil timing" | grep -v histogram | head -1
done
