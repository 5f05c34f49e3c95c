#!/bin/bash
# GPU box: counter passes over tools/placement_pmc.py; prints, per counter, the mean over the 4 launches into the
# slowest and into the fastest buffer (the last 8 launches of the render kernel in each pass).
ROOT=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$ROOT/gpurun_out/placement_pmc; rm -rf $OUT; mkdir -p $OUT; export TMPDIR=/tmp; cd $ROOT
i=0
IFS_OLD=$IFS
for set in "TCP_UTCL1_REQUEST_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_MISS_UNDER_MISS_sum" \
           "TCP_UTCL1_STALL_INFLIGHT_MAX_sum TCP_UTCL1_STALL_MULTI_MISS_sum TCP_UTCL1_SERIALIZATION_STALL_sum TCP_UTCL1_STALL_UTCL2_REQ_OUT_OF_CREDITS_sum" \
           "TCP_UTCL1_THRASHING_STALL_sum TCP_UTCL1_LFIFO_FULL_sum TCP_CLIENT_UTCL1_INFLIGHT_sum TCP_TCC_WRITE_REQ_HOLE_LATENCY" \
           "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_LEVEL_sum TCC_EA0_WRREQ_DRAM_CREDIT_STALL_sum TCC_TAG_STALL_sum"; do
  i=$((i+1))
  rocprofv3 --pmc $set --output-format csv -d $OUT/p$i -- python3 tools/placement_pmc.py > $OUT/o$i.log 2> $OUT/e$i.err || echo "pass $i failed"
  grep "times ms" $OUT/o$i.log
done
python3 - $OUT <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
for d in sorted(glob.glob(out + "/p*")):
    rows = []
    for f in glob.glob(d + "/*/*_counter_collection.csv"):
        rows += [r for r in csv.DictReader(open(f)) if "fe_env_kernel" in r["Kernel_Name"]]
    by = collections.defaultdict(list)
    for r in rows:
        by[r["Counter_Name"]].append((int(r["Dispatch_Id"]), float(r["Counter_Value"])))
    for name, v in sorted(by.items()):
        ids = sorted(set(i for i, _ in v))
        last8 = ids[-8:]
        per = collections.defaultdict(float)
        for i, x in v:
            per[i] += x          # (per-instance counters: summed per dispatch)
        slow = sum(per[i] for i in last8[:4]) / 4
        fast = sum(per[i] for i in last8[4:]) / 4
        print(f"{name:40s} slow {slow:16.1f}  fast {fast:16.1f}  slow/fast {slow / fast if fast else float('nan'):6.3f}")
        if name == "TCC_EA0_WRREQ_DRAM":  # per-instance spread
            inst = collections.defaultdict(lambda: [0.0, 0.0])
            for f in glob.glob(d + "/*/*_counter_collection.csv"):
                for r in csv.DictReader(open(f)):
                    if "fe_env_kernel" in r["Kernel_Name"] and r["Counter_Name"] == name and int(r["Dispatch_Id"]) in last8:
                        key = tuple((k, r[k]) for k in r if k.startswith("DIMENSION") or k in ("Instance", "Agent_Id"))
                        inst[key][0 if int(r["Dispatch_Id"]) in last8[:4] else 1] += float(r["Counter_Value"])
            vals = list(inst.values())
            if len(vals) > 1:
                s = [a for a, _ in vals]; f_ = [b for _, b in vals]
                print(f"   per instance ({len(vals)}): slow min/max {min(s):.0f}/{max(s):.0f}  fast min/max {min(f_):.0f}/{max(f_):.0f}")
PY
