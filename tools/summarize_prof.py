#!/usr/bin/env python3
"""Digest gpurun_out/prof_<tag>_c<cfg>/ (written by tools/profile_box.sh on the GPU box) into the tracked profiles/
directory: per-kernel stats, the step kernel split BY INSTANTIATION AND BY LAUNCH REGIME, HBM traffic per launch, a
short summary.

    python tools/summarize_prof.py <tag> <config>        (PROF_F32=1 for the f32-observation run)

Why the split: one bench.py run launches the step kernel from two places --
  * TRAINS: bench.KernelTrain issues trains of back-to-back launches straight through the C ABI (no Python work between
    them); this is what `roofline.kernel_ms` / `achieved` / `frac` are computed from.  They are the LAST
    3 x `kernel_launches_per_run` launches of the instantiation in the run (rounds 2 - 4), or -- round 5, `roofline.kernel_train_layout` --
    one train after each timed block (block, train, block, train ...: the tail of the instantiation's launches);
  * LOOP: the timed loop's env.step() launches (and its warm-up), one per Python iteration with trajectory slots that
    move every step; when the host keeps ahead of the GPU they queue back to back too, but every fence leaves the GPU
    idle and the first launches after a gap find the XCDs waking up staggered;
and rocprofv3's --stats table averages everything under one name.  The per-dispatch trace (*_kernel_trace.csv) tells them
apart; within each group a launch counts as "after a gap" when its predecessor on the queue ended more than GAP_NS
before it started.  `roofline.kernel` in the bench line names the instantiation; this script reports that row per group
and says which one the line's `kernel_ms` is to be compared with.
"""
import csv
import glob
import json
import os
import re
import statistics
import sys

GAP_NS = 4000  # a launch that starts <= 4 us after its predecessor ended was already queued behind it (boundary ~1.5 us)


def newest(pattern):
    return max(glob.glob(pattern), key=os.path.getmtime)


def short(name):
    name = name.replace("(anonymous namespace)::", "")
    name = re.sub(r"^void ", "", name)
    name = re.sub(r"\((anonymous namespace)?::?Params\)$|\(Params\)$", "", name)
    return name if len(name) < 110 else name[:107] + "..."


def is_step(name):
    return re.search(r"fe_env_kernel<[^>]*, (?:true|false), false, \d+>", name) is not None


def dispatches(trace_rows):
    """{instantiation: [(duration_ns, gap_to_predecessor_ns), ...] in dispatch order} for every step-kernel instantiation."""
    rows = sorted(trace_rows, key=lambda r: int(r["Start_Timestamp"]))
    out = {}
    prev_end = None
    for r in rows:
        name, st, en = r["Kernel_Name"], int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        if is_step(name):
            out.setdefault(short(name), []).append((en - st, st - prev_end if prev_end is not None else 1 << 60))
        prev_end = en
    return out


def stat(v):
    if not v:
        return None
    return {"calls": len(v), "avg_ns": sum(v) / len(v), "median_ns": statistics.median(v), "min_ns": min(v), "max_ns": max(v)}


def main():
    REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    tag, cfg = sys.argv[1], int(sys.argv[2])
    F32 = "_f32" if os.environ.get("PROF_F32") == "1" else ""
    src = os.path.join(REPO, "gpurun_out", f"prof_{tag}_c{cfg}{F32}")
    dst = os.path.join(REPO, "profiles")
    os.makedirs(dst, exist_ok=True)

    stats_csv = newest(os.path.join(src, "trace", "*", "*_kernel_stats.csv"))
    stats = list(csv.DictReader(open(stats_csv)))
    trace = list(csv.DictReader(open(stats_csv.replace("_kernel_stats.csv", "_kernel_trace.csv"))))
    with open(os.path.join(dst, f"{tag}_c{cfg}{F32}_kernel_stats.csv"), "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs", "StdDev"])
        for r in stats:
            w.writerow([short(r["Name"]), r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["Percentage"], r["MinNs"], r["MaxNs"], r["StdDev"]])
    # the run's full record (bench.py --detail; the stdout line is a summary of it since round 6)
    full = json.load(open(os.path.join(src, "bench_detail.json")))
    head = full["headline"]
    bench = {"roofline": head["roofline"], "dtype": full["dtype"], "steps": head["steps"], "warmup": head["warmup"], "value": head["value"],
             "ms_per_step": head["ms_per_step"], "repeats": head["repeats"],
             "config": {"workload": head["workload"], "eval_redraw": full["eval_redraw"], "launch": head["launch"]}}
    r = bench["roofline"]
    headline_kernel = r.get("kernel", "")
    by_form = dispatches(trace)
    # the instantiation the bench line names; fall back to the busiest step-kernel row for lines that predate `roofline.kernel`
    if headline_kernel not in by_form:
        headline_kernel = max(by_form, key=lambda k: len(by_form[k]))
    n_train = 3 * int(r.get("kernel_launches_per_run", 0))
    layout = r.get("kernel_train_layout")  # round 5: trains ALTERNATE with the timed blocks (block, train, block, train ...)

    def groups(name):
        d = by_form[name]
        if name == headline_kernel and layout:
            R, nb, nt = int(layout["repeats"]), int(layout["loop_launches_per_block"]), int(layout["train_launches"])
            tail = R * (nb + nt)
            if 0 < tail <= len(d):
                head, t = d[:len(d) - tail], d[len(d) - tail:]
                loop, trains = list(head), []
                for i in range(R):
                    loop += t[i * (nb + nt): i * (nb + nt) + nb]
                    trains += t[i * (nb + nt) + nb: (i + 1) * (nb + nt)]
                return {"loop": loop, "trains": trains}
        cut = len(d) - n_train if (name == headline_kernel and 0 < n_train < len(d)) else len(d)
        return {"loop": d[:cut], "trains": d[cut:]}

    with open(os.path.join(dst, f"{tag}_c{cfg}{F32}_step_kernel_regimes.csv"), "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["Instantiation", "Group", "Calls", "CallsAfterGap", "AverageNs", "MedianNs", "MinNs", "MaxNs", "AverageNsBackToBackOnly"])
        for name in sorted(by_form):
            for grp, d in groups(name).items():
                s_ = stat([x[0] for x in d])
                if s_:
                    b2b = [x[0] for x in d if x[1] <= GAP_NS]
                    w.writerow([name, grp, s_["calls"], sum(1 for x in d if x[1] > GAP_NS), f"{s_['avg_ns']:.0f}", f"{s_['median_ns']:.0f}",
                                s_["min_ns"], s_["max_ns"], f"{sum(b2b) / len(b2b):.0f}" if b2b else ""])
    g_head = groups(headline_kernel)
    tight, paced = stat([x[0] for x in g_head["trains"]]), stat([x[0] for x in g_head["loop"]])
    loop_gaps = sum(1 for x in g_head["loop"] if x[1] > GAP_NS)
    # the loop's launches in time order, in quarters: after the idle period of env construction the same kernel runs fast for
    # ~1 ms, then 10 - 20 % slower for a few ms, then settles (a clock / power transient of the box, visible at 64k envs where
    # the whole loop lasts 12 ms) -- the last quarter is the loop's settled figure
    q = len(g_head["loop"]) // 4
    quarters = [stat([x[0] for x in g_head["loop"][i * q:(i + 1) * q if i < 3 else None]]) for i in range(4)] if q >= 4 else []
    step_row = next(x for x in stats if short(x["Name"]) == headline_kernel)

    def pmc(kind):
        rows = list(csv.DictReader(open(newest(os.path.join(src, f"pmc_{kind}", "*", "*_counter_collection.csv")))))
        mine = [x for x in rows if short(x["Kernel_Name"]) == headline_kernel]
        if not mine:  # the counter passes run few steps: any step-kernel instantiation moves the same bytes
            mine = [x for x in rows if is_step(x["Kernel_Name"])]
        vals = [float(x["Counter_Value"]) for x in mine]
        return sum(vals) / len(vals), len(vals), mine[0]

    fetch_kb, nf, meta = pmc("fetch")
    write_kb, nw, _ = pmc("write")
    # MI355X_MICROARCH.md section HBM: FETCH_SIZE/WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE tallies 128-B
    # requests at 64 B, i.e. reads exactly 1/2 of a wide coalesced read stream -> doubled; WRITE_SIZE is exact
    # for 16-B/lane streaming stores.
    traffic = (2.0 * fetch_kb + write_kb) * 1024.0
    tj_path = os.path.join(dst, "hbm_traffic.json")
    tj = json.load(open(tj_path)) if os.path.exists(tj_path) else {}
    key = f"config{cfg}" + ("_f32" if bench["dtype"] == "f32" else "")
    tj[key] = {
        "workload": bench["config"]["workload"], "tag": tag,
        "kernel": headline_kernel,
        "fetch_size_kib_raw_per_launch": fetch_kb, "write_size_kib_per_launch": write_kb,
        "bytes_per_launch": traffic,
        "correction": "2*FETCH_SIZE + WRITE_SIZE, KiB->bytes (MI355X_MICROARCH.md HBM section); separate --pmc passes",
        "launches_averaged": {"fetch": nf, "write": nw},
        "rocprof_kernel_avg_ns": float(step_row["AverageNs"]), "rocprof_kernel_min_ns": float(step_row["MinNs"]),
        "rocprof_kernel_calls": int(step_row["Calls"]),
        "rocprof_trains_avg_ns": tight["avg_ns"] if tight else None, "rocprof_trains_calls": tight["calls"] if tight else 0,
        "rocprof_loop_avg_ns": paced["avg_ns"] if paced else None, "rocprof_loop_calls": paced["calls"] if paced else 0,
        "bench_kernel_interval_ns": r["kernel_ms"] * 1e6,
        "grid_size_threads": meta.get("Grid_Size"),
    }
    json.dump(tj, open(tj_path, "w"), indent=1, sort_keys=True)

    N, Bh, Bs = r["units_per_launch"], r["hbm_bytes_per_env_step"], r["survey_8d_bytes_per_env_step"]
    out_key = f"{tag}_c{cfg}" + ("_f32" if bench["dtype"] == "f32" else "")
    with open(os.path.join(dst, f"{out_key}_summary.md"), "w") as f:
        f.write(f"# {tag} config {cfg}: {bench['config']['workload']} ({bench['dtype']} observations)\n\n")
        f.write(f"command: `rocprofv3 --kernel-trace --stats --output-format csv -- python3 bench.py --config {cfg} --steps {bench['steps']} "
                f"--warmup {bench['warmup']} --no-cpu --no-extra --no-pmc --no-audition --repeats 2{' --obs-f32' if bench['dtype'] == 'f32' else ''}` "
                f"(+ separate `--pmc FETCH_SIZE` and `--pmc WRITE_SIZE` passes); `--no-audition`: every launch of the run writes the ring as allocated\n\n")
        form = headline_kernel.rstrip(">").split(",")[-1].strip()
        entry = {"0": "FORM 0 = the lean form (it writes the action copy too), `fe_env_step_traj` with `actions_out` only",
                 "1": "FORM 1 = the full form, `fe_env_step_traj`", "2": "FORM 2 = the lean form with the host flag of redraw='torch', `fe_env_step_traj_notify` with `actions_out` only",
                 "3": "FORM 3 = the full form with the host flag of redraw='torch', `fe_env_step_traj_notify`"}.get(form, f"FORM {form}")
        f.write(f"**The kernel the bench line times: `{headline_kernel}`** (`roofline.kernel`; {entry}: launched by the timed "
                f"loop's `env.step(..., rewards_out, dones_out, actions_out)` and by the C-ABI trains that alternate with its blocks; "
                f"eval_redraw = {bench['config'].get('eval_redraw')}).\n\n")
        f.write("| regime | calls | rocprof avg | median | min | max | HBM GB/s at avg | frac of 8 TB/s |\n|---|---|---|---|---|---|---|---|\n")
        for label, s_ in (("TRAINS (bench.KernelTrain: back-to-back C-ABI launches after each timed block; `roofline.kernel_ms` is measured here)", tight),
                          (f"LOOP (timed loop + warm-up: Python-issued env.step, moving trajectory slots; {loop_gaps} launches after an idle gap)", paced)):
            if s_:
                f.write(f"| {label} | {s_['calls']} | {s_['avg_ns']/1e3:.2f} us | {s_['median_ns']/1e3:.2f} us | {s_['min_ns']/1e3:.2f} us | "
                        f"{s_['max_ns']/1e3:.2f} us | {Bh*N/s_['avg_ns']:.0f} | {Bh*N/s_['avg_ns']/8000*100:.1f} % |\n")
        for i, s_ in enumerate(quarters):
            f.write(f"| LOOP, quarter {i + 1} of 4 in time order | {s_['calls']} | {s_['avg_ns']/1e3:.2f} us | {s_['median_ns']/1e3:.2f} us | {s_['min_ns']/1e3:.2f} us | "
                    f"{s_['max_ns']/1e3:.2f} us | {Bh*N/s_['avg_ns']:.0f} | {Bh*N/s_['avg_ns']/8000*100:.1f} % |\n")
        f.write(f"| all launches of this instantiation (rocprofv3 --stats row) | {step_row['Calls']} | {float(step_row['AverageNs'])/1e3:.2f} us | | "
                f"{float(step_row['MinNs'])/1e3:.2f} us | {float(step_row['MaxNs'])/1e3:.2f} us | {Bh*N/float(step_row['AverageNs']):.0f} | "
                f"{Bh*N/float(step_row['AverageNs'])/8000*100:.1f} % |\n\n")
        if quarters and paced and tight and paced["avg_ns"] > 1.03 * tight["avg_ns"]:
            f.write("The LOOP average is above the TRAINS average although almost all of its launches are queued back to back: in time order "
                    "(quarters above) the same kernel runs at the TRAINS figure at first, 10 - 20 % longer from ~1 ms after the GPU left its idle state, "
                    "and comes back over ~10 ms -- a clock / power transient after the idle period of env construction, not a property of the "
                    "launch path (moving trajectory slots cost +0.2 us, NOTES.md round 4).  A run with more blocks (the default command: "
                    "median of up to 40) sits in the settled regime.\n\n")
        if tight:
            dev = (r["kernel_ms"] * 1e6 - tight["avg_ns"]) / tight["avg_ns"] * 100
            f.write(f"* bench.py's own figure in this run: HIP-event launch interval {r['kernel_ms']*1e3:.2f} us over trains of "
                    f"{r.get('kernel_launches_per_run', '?')} launches = rocprof TRAINS average {tight['avg_ns']/1e3:.2f} us "
                    f"{dev:+.1f} % (the interval includes the ~1.5 us launch boundary; the kernel's own duration does not)\n")
        f.write(f"* bench.py under the profiler: {bench['value']:.4g} env-steps/s, {bench['ms_per_step']*1e3:.2f} us/step wall (median of "
                f"{bench['repeats'][next(iter(bench['repeats']))]['blocks']} blocks; the wall step is the LOOP launches + the fences around each block)\n")
        others = [f"`{n}` {len(v)} calls, avg {sum(x[0] for x in v) / len(v) / 1e3:.2f} us" for n, v in sorted(by_form.items()) if n != headline_kernel]
        f.write("* other step-kernel instantiations in this run: " + ("; ".join(others) if others else "none") + "\n")
        f.write(f"* HBM bytes that must move per launch (observation write + state + outputs): {Bh} B x {N} envs = {Bh*N/1e6:.1f} MB "
                f"(this is what `roofline.achieved` / `frac` divide by the launch duration)\n")
        avg = tight["avg_ns"] if tight else float(step_row["AverageNs"])
        f.write(f"* PMC (per launch): FETCH_SIZE {fetch_kb:.1f} KiB raw, WRITE_SIZE {write_kb:.1f} KiB -> HBM traffic ~ {traffic/1e6:.1f} MB "
                f"({traffic/avg:.0f} GB/s at the TRAINS average); traffic / compulsory bytes = {traffic/(Bh*N):.3f}\n")
        f.write(f"* SURVEY 8(d) formula incl. the L2-served window re-read: {Bs} B per env-step = {Bs*N/1e6:.1f} MB per launch; the "
                f"{r['l2_read_bytes_per_env_step']} B window part is L2 / Infinity-Cache traffic ({r['l2_read_bytes_per_env_step']*N/avg:.0f} GB/s), not HBM\n")
        f.write(f"* grid {meta.get('Grid_Size')} threads of 256 ({bench['config']['launch']})\n")
    print(open(os.path.join(dst, f"{out_key}_summary.md")).read())


if __name__ == "__main__":
    main()
