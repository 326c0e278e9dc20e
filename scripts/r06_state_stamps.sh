# Round 6: phase stamps of the Tangent walk (state_pencil_k, CH p=2 128^3, a steady-state launch; -DIGX_DEBUG build)
export IGX_USE_DEBUG_LIB=1 IGX_DEBUG_TIMING=12
for nf in 0 2; do
  echo "== CH 128 Tangent IGX_DEBUG_NOFLUSH=$nf"
  IGX_DEBUG_NOFLUSH=$nf python bench.py --form cahnhilliard --size 128 --two-calls --steps 3 --warmup 2 --no-cpu-baseline --no-live-traffic 2>&1 >/dev/null | grep "igx pencil timing" | grep -v histogram | head -1
done
