#!/bin/bash
# GPU box: rocprofv3 evidence for the MLP-head rollout kernel: kernel-trace statistics + MFMA counters.
FORM=${1:-mlp64}
ROOT=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$ROOT/gpurun_out/mlp_prof; rm -rf $OUT; mkdir -p $OUT; export TMPDIR=/tmp; cd $ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 tools/fused_bench.py 2 $FORM > $OUT/trace.log 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES SQ_BUSY_CYCLES SQ_INSTS_MFMA SQ_ACTIVE_INST_VALU SQ_INSTS_VALU GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc -- python3 tools/fused_bench.py 2 $FORM > $OUT/pmc.log 2>&1 || echo "pmc pass failed"
python3 - $OUT <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
for f in glob.glob(out + "/trace/*/*_kernel_stats.csv"):
    for r in csv.DictReader(open(f)):
        if "mlp" in r["Name"]:
            print(f"kernel-trace: {r['Name'][:70]} calls {r['Calls']} avg {float(r['AverageNs'])/1e3:.1f} us per launch of 32 steps -> {float(r['AverageNs'])/32e3:.2f} us/step")
agg = collections.defaultdict(list)
for f in glob.glob(out + "/pmc/*/*_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "mlp" in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in sorted(agg.items()):
    print(f"pmc {k:32s} n={len(v)} mean={sum(v)/len(v):.6g}")
PY
