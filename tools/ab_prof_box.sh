#!/bin/bash
# GPU box: rocprofv3 kernel-trace statistics of two builds on the SAME box, alternating A B A B.
# usage: tools/ab_prof_box.sh <config> <armA> <armB> [steps]
CFG=${1:-2}; A=${2:-r01like}; B=${3:-product}; K=${4:-400}
ROOT=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$ROOT/gpurun_out/abprof_c$CFG; rm -rf $OUT; mkdir -p $OUT; export TMPDIR=/tmp; cd $ROOT
for i in 1 2; do for ARM in $A $B; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/$ARM$i -- python3 tools/run_arm.py $CFG $ARM $K > $OUT/$ARM$i.log 2>&1 || { echo "$ARM$i failed"; tail -3 $OUT/$ARM$i.log; }
  python3 - $OUT/$ARM$i $ARM$i <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/*/*_kernel_stats.csv")[0]
rows = [r for r in csv.DictReader(open(f)) if "fe_env_kernel" in r["Name"]]
r = max(rows, key=lambda r: int(r["Calls"]))
print(f"{sys.argv[2]:12s} {r['Name'][:60]:60s} calls {r['Calls']:>5s}  avg {float(r['AverageNs'])/1e3:10.2f} us  min {float(r['MinNs'])/1e3:10.2f}  max {float(r['MaxNs'])/1e3:10.2f}", flush=True)
PY
done; done
