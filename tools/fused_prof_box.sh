#!/bin/bash
# GPU box: rocprofv3 evidence for a fused-rollout kernel (MLP / LSTM head): kernel-trace statistics + MFMA / VALU counters.
#   tools/fused_prof_box.sh <form> [W] [K]        e.g.  mlp64      or      lstm128 4 8
FORM=${1:-mlp64}; export FUSED_K=${3:-32}
if [ -n "$2" ]; then export FUSED_W=$2; fi
ROOT=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$ROOT/gpurun_out/fused_prof_$FORM; rm -rf $OUT; mkdir -p $OUT; export TMPDIR=/tmp; cd $ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 tools/fused_bench.py 2 $FORM > $OUT/trace.log 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES SQ_BUSY_CYCLES SQ_INSTS_MFMA SQ_ACTIVE_INST_VALU SQ_INSTS_VALU GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc -- python3 tools/fused_bench.py 2 $FORM > $OUT/pmc.log 2>&1 || echo "pmc pass failed"
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS --output-format csv -d $OUT/pmc2 -- python3 tools/fused_bench.py 2 $FORM > $OUT/pmc2.log 2>&1 || echo "pmc2 pass failed"
python3 - $OUT $FUSED_K <<'PY'
import csv, glob, sys, collections
out, K = sys.argv[1], int(sys.argv[2])
for f in glob.glob(out + "/trace/*/*_kernel_stats.csv"):
    for r in csv.DictReader(open(f)):
        if "rollout" in r["Name"]:
            print(f"kernel-trace: {r['Name'][:80]} calls {r['Calls']} avg {float(r['AverageNs'])/1e3:.1f} us per launch of {K} steps -> {float(r['AverageNs'])/(K*1e3):.2f} us/step")
for sub in ("pmc", "pmc2"):
    agg = collections.defaultdict(list)
    for f in glob.glob(out + f"/{sub}/*/*_counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            if "rollout" in r["Kernel_Name"]:
                agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in sorted(agg.items()):
        print(f"pmc {k:32s} n={len(v)} mean={sum(v)/len(v):.6g}")
PY
grep "config" $OUT/trace.log
