#!/usr/bin/env bash
# usage: sweep_env.sh VAR "v1 v2 ..." -- bench args : one bench.py line per value (value, ms_per_step, dominant kernel ms, shader clock)
var=$1; vals=$2; shift 3
for v in $vals; do
  env $var=$v python bench.py "$@" --no-cpu-baseline --no-live-traffic 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); r=d['roofline']
print('$var=$v', 'ms/step %.2f' % d['ms_per_step'], 'value %.3e' % d['value'], 'launch ms %.3f' % r['avg_launch_ms'], 'frac %.3f' % r['frac'], 'MHz', r['shader_clock_mhz'], 'frac@clock %.3f' % (r['frac_at_measured_clock'] or 0), d['per_step']['ms'])"
done
