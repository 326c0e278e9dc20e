# Round 6: phase stamps of the fused IFunction + IJacobian walk against the Tangent alone (debug build), CH p=2 128^3, a steady-state launch.
# IGX_DEBUG_NOFLUSH=2: the columns split the flush phase instead: "mfma" = the Residual's leave, "wait" = the band row's leave, "flush" = the next element's state
export IGX_USE_DEBUG_LIB=1 IGX_DEBUG_TIMING=12
for nf in 0 2; do for tc in "" "--two-calls"; do
  echo "== CH 128 ${tc:-fused} IGX_DEBUG_NOFLUSH=$nf"
  IGX_DEBUG_NOFLUSH=$nf python bench.py --form cahnhilliard --size 128 $tc --steps 3 --warmup 2 --no-cpu-baseline --no-live-traffic 2>&1 >/dev/null | grep "igx pencil timing" | grep -v histogram | head -1
done; done
