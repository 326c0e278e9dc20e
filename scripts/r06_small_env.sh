# Round 6: the step model of the plain Gram walk across mesh sizes (p = 3), against the previous choice (IGX_SMALL_WPB=0)
for w in 0 1; do echo "== IGX_SMALL_WPB=$w"; IGX_SMALL_WPB=$w REPS=10 SIZES="16 24 32 48 64 96 128" python3 scripts/r06_small.py 2>&1 | grep "p=3"; done
