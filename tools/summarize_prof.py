#!/usr/bin/env python3
"""Digest gpurun_out/prof_<tag>_c<cfg>/ (written by tools/profile_box.sh on the GPU box) into
the tracked profiles/ directory: per-kernel stats, HBM traffic per launch, a short summary."""
import csv, glob, json, os, sys

def newest(pattern):
    return max(glob.glob(pattern), key=os.path.getmtime)

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag, cfg = sys.argv[1], int(sys.argv[2])
src = os.path.join(REPO, "gpurun_out", f"prof_{tag}_c{cfg}" + ("_f32" if os.environ.get("PROF_F32") == "1" else ""))
dst = os.path.join(REPO, "profiles")
os.makedirs(dst, exist_ok=True)

def short(name):
    name = name.replace("(anonymous namespace)::", "")
    return name if len(name) < 110 else name[:107] + "..."

stats = list(csv.DictReader(open(newest(os.path.join(src, "trace", "*", "*_kernel_stats.csv")))))
F32 = "_f32" if os.environ.get("PROF_F32") == "1" else ""
src_note = None
with open(os.path.join(dst, f"{tag}_c{cfg}{F32}_kernel_stats.csv"), "w", newline="") as f:
    w = csv.writer(f)
    w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs", "StdDev"])
    for r in stats:
        w.writerow([short(r["Name"]), r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["Percentage"], r["MinNs"], r["MaxNs"], r["StdDev"]])
# robust pick: the fe_env_kernel row with the most calls is the step kernel
step = max((r for r in stats if "fe_env_kernel" in r["Name"]), key=lambda r: int(r["Calls"]))

def pmc(kind):
    rows = list(csv.DictReader(open(newest(os.path.join(src, f"pmc_{kind}", "*", "*_counter_collection.csv")))))
    vals = [float(r["Counter_Value"]) for r in rows if "fe_env_kernel" in r["Kernel_Name"] and r["Kernel_Name"] == step["Name"]]
    meta = next(r for r in rows if r["Kernel_Name"] == step["Name"])
    return sum(vals) / len(vals), len(vals), meta

fetch_kb, nf, meta = pmc("fetch")
write_kb, nw, _ = pmc("write")
bench = json.loads(open(os.path.join(src, "bench_trace.json")).read().strip().splitlines()[-1])
# MI355X_MICROARCH.md section HBM: FETCH_SIZE/WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE tallies 128-B
# requests at 64 B, i.e. reads exactly 1/2 of a wide coalesced read stream -> doubled; WRITE_SIZE is exact
# for 16-B/lane streaming stores.
traffic = (2.0 * fetch_kb + write_kb) * 1024.0
tj_path = os.path.join(dst, "hbm_traffic.json")
tj = json.load(open(tj_path)) if os.path.exists(tj_path) else {}
tj[f"config{cfg}" + ("_f32" if bench["dtype"] == "f32" else "")] = {
    "workload": bench["config"]["workload"], "tag": tag,
    "kernel": short(step["Name"]),
    "fetch_size_kib_raw_per_launch": fetch_kb, "write_size_kib_per_launch": write_kb,
    "bytes_per_launch": traffic,
    "correction": "2*FETCH_SIZE + WRITE_SIZE, KiB->bytes (MI355X_MICROARCH.md HBM section); separate --pmc passes",
    "launches_averaged": {"fetch": nf, "write": nw},
    "rocprof_kernel_avg_ns": float(step["AverageNs"]), "rocprof_kernel_min_ns": float(step["MinNs"]),
    "rocprof_kernel_calls": int(step["Calls"]),
    "grid_size_threads": meta.get("Grid_Size"),
}
json.dump(tj, open(tj_path, "w"), indent=1, sort_keys=True)
r = bench["roofline"]
N, Bh, Bs = r["units_per_launch"], r["hbm_bytes_per_env_step"], r["survey_8d_bytes_per_env_step"]
avg = float(step["AverageNs"])
key = f"{tag}_c{cfg}" + ("_f32" if bench["dtype"] == "f32" else "")
with open(os.path.join(dst, f"{key}_summary.md"), "w") as f:
    f.write(f"# {tag} config {cfg}: {bench['config']['workload']} ({bench['dtype']} observations)\n\n")
    f.write(f"command: `rocprofv3 --kernel-trace --stats --output-format csv -- python3 bench.py --config {cfg} --steps {bench['steps']} --warmup {bench['warmup']} --no-cpu --no-extra --no-audition{' --obs-f32' if bench['dtype'] == 'f32' else ''}` (+ separate `--pmc FETCH_SIZE` and `--pmc WRITE_SIZE` passes)\n\n")
    f.write(f"* step kernel `{short(step['Name'])}`: {step['Calls']} calls, avg {avg/1e3:.2f} us, min {float(step['MinNs'])/1e3:.2f} us, max {float(step['MaxNs'])/1e3:.2f} us ({step['Percentage']} % of GPU time)\n")
    f.write(f"* bench.py under the profiler: {bench['value']:.4g} env-steps/s, {bench['ms_per_step']*1e3:.2f} us/step wall (median of {bench['repeats']['single_gpu']['blocks']} blocks), HIP-event average launch interval {r['kernel_ms']*1e3:.2f} us (kernel + launch boundary, tight C-ABI loop)\n")
    f.write(f"* HBM bytes that must move per launch (observation write + state + outputs): {Bh} B x {N} envs = {Bh*N/1e6:.1f} MB -> {Bh*N/avg:.0f} GB/s at the rocprof average = **{Bh*N/avg/8000*100:.1f} % of 8 TB/s** (this is bench.py's `roofline.achieved` / `frac`)\n")
    f.write(f"* PMC (per launch): FETCH_SIZE {fetch_kb:.1f} KiB raw, WRITE_SIZE {write_kb:.1f} KiB -> HBM traffic ~ {traffic/1e6:.1f} MB ({traffic/avg:.0f} GB/s at the rocprof average); traffic / compulsory bytes = {traffic/(Bh*N):.3f}\n")
    f.write(f"* SURVEY 8(d) formula incl. the L2-served window re-read: {Bs} B per env-step = {Bs*N/1e6:.1f} MB per launch; the {r['l2_read_bytes_per_env_step']} B window part is L2 / Infinity-Cache traffic ({r['l2_read_bytes_per_env_step']*N/avg:.0f} GB/s), not HBM\n")
    f.write(f"* grid {meta.get('Grid_Size')} threads of 256 ({bench['config']['launch']})\n")
print(open(os.path.join(dst, f"{key}_summary.md")).read())
