# Round 6: band_pt (config 5's share of one GPU: NS-VMS p=3 96^3 on the NURBS net), schedule experiments in one lease.
# IGX_BAND_PRIO=k: s_setprio 2 through a workgroup's first k layers; IGX_BAND_RMW_PRIO=1: s_setprio 3 through a read-add-write; IGX_NSEG: segments per pencil (24 = four layers)
run() { echo -n "$1: "; env $1 python bench.py --form nsvms --steps 4 --warmup 1 --no-cpu-baseline --no-live-traffic 2>/dev/null | python -c "
import sys,json; l=json.loads(sys.stdin.read()); r=l['roofline']; print(round(l['value']/1e6,3), 'M el/s  ms/step', round(l['ms_per_step'],2), ' band_pt launch ms', round(r['avg_launch_ms'],3), 'frac', round(r['frac'],3))"; }
run "IGX_BAND_PRIO=0"
run "IGX_BAND_PRIO=1"
run "IGX_BAND_PRIO=2"
run "IGX_BAND_RMW_PRIO=1"
run "IGX_BAND_PRIO=1 IGX_BAND_RMW_PRIO=1"
run "IGX_NSEG=16"
run "IGX_NSEG=12"
run "IGX_NSEG=16 IGX_BAND_PRIO=1"
run "IGX_NSEG=32"
