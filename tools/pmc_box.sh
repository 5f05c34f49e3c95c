#!/bin/bash
# GPU box: extra PMC passes for the step kernel of one config (diagnostics, not the roofline numbers).
CFG=${1:-3}; TAG=${2:-diag}
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/pmc_${TAG}_c${CFG}
mkdir -p $OUT; export TMPDIR=/tmp; cd $ROOT
i=0
for set in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" \
           "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VMEM" \
           "TCC_HIT_sum TCC_MISS_sum TCC_EA0_WRREQ_sum TCC_EA0_RDREQ_sum" \
           "GRBM_GUI_ACTIVE GRBM_COUNT"; do
  i=$((i+1))
  rocprofv3 --pmc $set --output-format csv -d $OUT/p$i -- python3 bench.py --config $CFG --steps 8 --warmup 4 --no-cpu --no-extra --no-pmc --repeats 2 > $OUT/b$i.json 2> $OUT/e$i.err || echo "pass $i failed"
done
python3 - <<PY
import csv, glob, collections, re
agg = collections.defaultdict(list)
for f in glob.glob("$OUT/p*/*/*_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if re.search(r"fe_env_kernel<[^>]*, (?:true|false), false, \d+>", r["Kernel_Name"]):  # the step kernel (RESET_ONLY = false)
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in sorted(agg.items()):
    print(f"{k:28s} n={len(v):3d} mean={sum(v)/len(v):.6g}")
PY
