#!/bin/bash
# GPU box: rocprofv3 kernel-trace statistics of the DEFAULT bench command (headline + extra_configs legs), so that
# bench.py's own HIP-event kernel figures can be checked against the profiler's averages of the same run.
TAG=${1:-r02}
ROOT=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$ROOT/gpurun_out/prof_${TAG}_default; rm -rf $OUT; mkdir -p $OUT; export TMPDIR=/tmp; cd $ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-pmc > $OUT/bench.json 2> $OUT/bench.err || exit 1
python3 - $OUT $TAG <<'PY'
import csv, glob, json, os, sys
out, tag = sys.argv[1], sys.argv[2]
stats = list(csv.DictReader(open(glob.glob(out + "/trace/*/*_kernel_stats.csv")[0])))
b = json.loads(open(out + "/bench.json").read().strip().splitlines()[-1])
lines = [f"# {tag}: `rocprofv3 --kernel-trace --stats -- python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-pmc`", ""]
lines.append(f"bench.py line of the same run: value {b['value']:.4g} env-steps/s, ms_per_step {b['ms_per_step']*1e3:.2f} us, kernel_ms {b['roofline']['kernel_ms']*1e3:.2f} us (HIP events), frac {b['roofline']['frac']:.3f}")
lines.append(f"(the line itself: {len(open(out + '/bench.json').read().strip().splitlines()[-1])} characters)")
for e in b["extra_configs"]:
    lines.append(f"  extra: {e['workload']}: value {e['value']:.4g}, kernel_ms {e['roofline']['kernel_ms']*1e3:.1f} us, frac {e['roofline']['frac']:.3f}")
for x in b.get("legs", []):
    lines.append(f"  leg: {x['leg']}: value {x.get('value')}, ms_per_step {x.get('ms_per_step')}, frac {x.get('frac')}")
lines += ["", "| kernel | calls | avg us | min us | max us | % of GPU time |", "|---|---|---|---|---|---|"]
for r in sorted(stats, key=lambda r: -float(r["TotalDurationNs"]))[:8]:
    lines.append(f"| `{r['Name'].replace('(anonymous namespace)::', '')[:70]}` | {r['Calls']} | {float(r['AverageNs'])/1e3:.2f} | {float(r['MinNs'])/1e3:.2f} | {float(r['MaxNs'])/1e3:.2f} | {r['Percentage']} |")
lines += ["", "(`fe_env_kernel<double, 2, true, false, 2>` is the config-2 step kernel as the headline's loop and trains launch it (redraw='torch' with rewards / dones / action copy in trajectory slots: the lean + host-flag form; also the `reference_semantics` leg); `..., 0>` = the lean form without the flag (the `device_redraw` leg, the hipGraph strong-scaling legs, `two_streams`); `..., 1>` / `..., 3>` = the full forms (descriptors / statistics / evaluate mode: not launched by this command).  The single-asset instantiations also serve the strong-scaling legs at 8 192 - 65 536 envs, and `<double, 2, false, false, *>` serves configs 3, 4 and 5 in the same process, so their averages mix launches of very different sizes: per-config figures are in `profiles/<tag>_c<cfg>_summary.md`.)"]
with open(os.path.join(out, f"{tag}_default_cmd_summary.md"), "w") as f:  # gpurun merges gpurun_out/ back: copy it to profiles/ from there
    f.write("\n".join(lines) + "\n")
print("\n".join(lines))
PY
