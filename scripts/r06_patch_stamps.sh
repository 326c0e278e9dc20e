# Round 6: cycle stamps and the per-CU timeline of a steady-state launch of the patch walk (config 2: Poisson p=2 128^3 System; -DIGX_DEBUG build)
export IGX_USE_DEBUG_LIB=1
for n in 6 7; do IGX_DEBUG_TIMING=$n python bench.py --form poisson --degree 2 --size 128 --steps 3 --warmup 2 --no-cpu-baseline --no-live-traffic 2>&1 >/dev/null | grep "igx patch timing"; done
