#!/bin/bash
# GPU box: A/B the cache policy of the observation stores (plain vs sc1 / nt / sc0|sc1 buffer stores):
# kernel time + FETCH_SIZE (does the table stay in L2 when the obs stream does not allocate there?).
ROOT=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$ROOT/gpurun_out/store_policy; mkdir -p $OUT; export TMPDIR=/tmp; cd $ROOT
for AUX in -1 16 2 17 -1 16; do
  FE_STORE_AUX=$AUX python3 -m finenvs_amd.csrc.build --force > /dev/null 2>&1 || { echo "build failed aux=$AUX"; continue; }
  python3 -m pytest tests/test_hip_parity.py -m gpu -q -x -k "seeded and (1024 or 333)" 2>&1 | tail -1
  for CFG in "$@"; do
    python3 bench.py --config $CFG --steps 48 --warmup 16 --no-cpu 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('aux=$AUX cfg=$CFG kernel %.1f us  value %.4g' % (d['roofline']['kernel_ms']*1e3, d['value']), flush=True)"
    rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/p -- python3 bench.py --config $CFG --steps 8 --warmup 4 --no-cpu > /dev/null 2>&1
    python3 - $OUT/p <<'PY'
import csv, glob, sys
f = max(glob.glob(sys.argv[1] + "/*/*_counter_collection.csv"), key=len)
v = [float(r["Counter_Value"]) for r in csv.DictReader(open(f)) if "fe_env_kernel" in r["Kernel_Name"] and "false>(" in r["Kernel_Name"]]
print(f"      FETCH_SIZE {sum(v)/len(v)/1024:.1f} MiB raw per launch", flush=True)
PY
    rm -rf $OUT/p
  done
done
python3 -m finenvs_amd.csrc.build --force > /dev/null 2>&1
