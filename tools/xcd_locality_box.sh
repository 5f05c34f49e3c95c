#!/bin/bash
# GPU box: does keeping tile % 8 (the XCD label) constant per workgroup matter?  Compare a grid that is a
# multiple of 8 with one that is not (tiles then wander over XCDs), by FETCH_SIZE and kernel time.
CFG=${1:-3}
ROOT=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$ROOT/gpurun_out/xcd_c${CFG}; mkdir -p $OUT; export TMPDIR=/tmp; cd $ROOT
for G in 1536 1533 1536 1533; do
  FE_GRID=$G rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/g$G -- python3 bench.py --config $CFG --steps 16 --warmup 8 --no-cpu > $OUT/b$G.json 2> $OUT/e$G.err
  python3 - $OUT/g$G $G $OUT/b$G.json <<'PY'
import csv, glob, json, sys
f = max(glob.glob(sys.argv[1] + "/*/*_counter_collection.csv"), key=len)
v = [float(r["Counter_Value"]) for r in csv.DictReader(open(f)) if "fe_env_kernel" in r["Kernel_Name"] and "false>(" in r["Kernel_Name"]]
b = json.loads(open(sys.argv[3]).read().strip().splitlines()[-1])
print(f"grid {sys.argv[2]}: FETCH_SIZE {sum(v)/len(v)/1024:.1f} MiB raw per launch over {len(v)} launches; kernel {b['roofline']['kernel_ms']*1e3:.1f} us", flush=True)
PY
  rm -rf $OUT/g$G
done
