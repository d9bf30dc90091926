"""Collect the rocprofv3 evidence bench.py's roofline object refers to (run ON the GPU box, from the repo root):

    python tools/collect_profiles.py <tag>

1. `rocprofv3 --kernel-trace --stats` of `python bench.py --steps 100 --warmup 10 --no-cpu-baseline` (bench.py's defaults)
   -> gpurun_out/rocprof_<tag>_kernel_stats.txt (per-kernel totals / averages from the top_kernels view)
2. two separate `rocprofv3 --pmc <C> --kernel-trace --output-format csv` passes (C = FETCH_SIZE, WRITE_SIZE; counters are
   collected in their own runs, without any other trace domain) of the same command with fewer steps
   -> gpurun_out/pmc_<tag>_hbm_traffic.{txt,json}: HBM bytes per launch of the step kernel.  FETCH_SIZE / WRITE_SIZE are
   in KiB; on gfx950 FETCH_SIZE counts 64 B per wide (128 B) read request, so reads are doubled (MI355X_MICROARCH.md,
   HBM / rocprofv3 section); both the corrected and the raw figure are reported.
Copy the summaries you want judged into profiles/.
"""
import csv
import glob
import json
import os
import sqlite3
import subprocess
import sys

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
OUT = os.path.join(ROOT, "gpurun_out")
sys.path.insert(0, ROOT)
from bench import WORKLOADS  # noqa: E402  (kernel names, worlds per GPU and algorithmic bytes per env-step of every bench workload)

KERNEL = "grx_fetch_step_kernel"
ALGO_BYTES = 715 * 4096


def _build_id():
    """device-code identity of the library the profiled command loads (gymnasium_robotics_amd._native.build_id): bench.py attaches a summary only to that build"""
    from gymnasium_robotics_amd import _native

    return _native.build_id()


def _git_head():
    try:      # (the GPU box's snapshot has no .git: tools/final_bench.sh passes the hash of the tree it was cut from)
        return os.environ.get("GRX_GIT_HEAD") or subprocess.run(["git", "-C", ROOT, "rev-parse", "--short", "HEAD"], capture_output=True, text=True).stdout.strip() or None
    except Exception:
        return None


def _whole_batch(grid_threads, full_grid):
    """a launch over ALL worlds: one workgroup per world, or (the Fetch split step, include/grx_capi.h grx_fetch_buffers.split_parts) 2 - 8 workgroups per world"""
    return grid_threads > 0 and grid_threads % full_grid == 0 and grid_threads // full_grid <= 8


def run(cmd, log):
    env = dict(os.environ, TMPDIR="/tmp")
    with open(log, "w") as f:
        subprocess.run(cmd, cwd="/tmp", env=env, stdout=f, stderr=subprocess.STDOUT, check=False)


def kernel_stats(tag, workload=None):
    """workload: one of bench.py's --workload names (antmaze, hand_touch, hand_reach): kernel stats of that BASELINE config instead of cfg 2"""
    suffix = f"_{workload}" if workload else ""
    d = os.path.join(OUT, f"rocprof_{tag}{suffix}")
    extra = (["--workload", workload] if workload else []) + os.environ.get("GRX_COLLECT_STATS_EXTRA", "").split()      # (e.g. "--stages 2": rename the summary afterwards)
    cmd = ["rocprofv3", "--kernel-trace", "--stats", "-d", d, "-o", "bench", "--", sys.executable, os.path.join(ROOT, "bench.py"),
           "--steps", "100", "--warmup", "10", "--no-cpu-baseline"] + extra     # bench.py's own defaults: the averages are over the same launches
    run(cmd, os.path.join(OUT, f"rocprof_{tag}{suffix}.log"))
    dbs = glob.glob(os.path.join(d, "**", "*.db"), recursive=True)
    lines = [f"rocprofv3 --kernel-trace --stats -- python bench.py --steps 100 --warmup 10 --no-cpu-baseline {' '.join(extra)}  (MI355X, build '{tag}')",
             "source: top_kernels view of the rocprofv3 results database; durations in us",
             "name | total_calls | total_duration_us | average_us | percentage"]
    if dbs:
        con = sqlite3.connect(dbs[0])
        for name, calls, total, avg, pct in con.execute("select name, total_calls, total_duration, average, percentage from top_kernels"):
            short = name if len(name) < 160 else name[:157] + "..."
            lines.append(f"{short} | {calls} | {total:.0f} | {avg:.1f} | {pct:.3f}")
    else:
        lines.append("(no results database produced -- see the log)")
    with open(os.path.join(OUT, f"rocprof_{tag}_kernel_stats{suffix}.txt"), "w") as f:
        f.write("\n".join(lines) + "\n")
    print("\n".join(lines[:6]))


def pmc(tag, workload=None):
    """HBM traffic per launch of the workload's step kernel: two separate --pmc passes (FETCH_SIZE, WRITE_SIZE), nothing else traced"""
    w = WORKLOADS[workload or "fetch"]
    kernel, algo_bytes = w["kernel"], w["algo"] * w["worlds"]
    suffix = f"_{workload}" if workload else ""
    extra = (["--workload", workload] if workload else []) + os.environ.get("GRX_COLLECT_EXTRA", "").split()      # (A/B runs: e.g. "--preroll 0")
    res, meta = {}, {}
    for cnt in ("FETCH_SIZE", "WRITE_SIZE"):
        d = os.path.join(OUT, f"pmc_{tag}_{cnt}{suffix}")
        # --no-stagger: episodes in lock-step, so the 15 steps of this run contain NO reset (the masked reset-time forward launches and the hand families'
        # settle chains run the same kernel on a handful of worlds and used to be averaged in: round 2 reported a "traffic" BELOW the algorithmic bytes)
        cmd = ["rocprofv3", "--pmc", cnt, "--kernel-trace", "--output-format", "csv", "-d", d, "-o", "pmc", "--", sys.executable,
               os.path.join(ROOT, "bench.py"), "--steps", "12", "--warmup", "3", "--no-cpu-baseline", "--no-stagger"] + extra
        run(cmd, os.path.join(OUT, f"pmc_{tag}_{cnt}{suffix}.log"))
        vals, dropped = [], 0
        full_grid = ((w["worlds"] + 7) // 8) * 8 * 64      # threads of a launch over all worlds (grids are rounded up to a multiple of 8 workgroups)
        for path in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            with open(path) as f:
                for row in csv.DictReader(f):
                    if kernel in row.get("Kernel_Name", "") and row.get("Counter_Name") == cnt:
                        if not _whole_batch(int(float(row.get("Grid_Size") or 0)), full_grid):      # a compacted side launch (settle chain, compacted reset): not a step of the whole batch
                            dropped += 1
                            continue
                        vals.append(float(row["Counter_Value"]))
                        meta = {k: row.get(k) for k in ("Grid_Size", "Workgroup_Size", "LDS_Block_Size", "VGPR_Count", "Accum_VGPR_Count", "SGPR_Count", "Scratch_Size")}
        # the first launches of the run are the reset() of all worlds (forward-only over the full grid) and the warm-up: keep the timed region's step launches
        vals = vals[-12:]
        meta["launches_dropped_by_grid_size"] = dropped
        res[cnt] = vals
    lines = [f"rocprofv3 --pmc <C> --kernel-trace --output-format csv -- python bench.py --steps 12 --warmup 3 --no-cpu-baseline --no-stagger {' '.join(extra)}  (separate passes, MI355X, build '{tag}'; "
             f"the last 12 full-grid launches of the step kernel = the timed region, no reset launches among them)",
             f"kernel dispatch info: {meta}"]
    summary = {}
    for cnt, vals in res.items():
        if vals:
            mean = sum(vals) / len(vals)
            summary[cnt] = mean
            lines.append(f"{cnt}: {len(vals)} launches of {kernel}: mean {mean:.1f} KiB  min {min(vals):.1f}  max {max(vals):.1f}")
        else:
            lines.append(f"{cnt}: no samples (see the log)")
    out = {"kernel": kernel, "build": tag, "build_id": _build_id(), "git": _git_head(), "algorithmic_bytes_per_launch": algo_bytes}
    if len(summary) == 2:
        fetch, write = summary["FETCH_SIZE"] * 1024, summary["WRITE_SIZE"] * 1024
        traffic, raw = 2 * fetch + write, fetch + write
        lines.append(f"per launch: FETCH_SIZE {fetch/1e3:.0f} kB (x2 gfx950 wide-read correction: {2*fetch/1e3:.0f} kB), WRITE_SIZE {write/1e3:.0f} kB; "
                     f"algorithmic bytes {w['algo']} B x {w['worlds']} worlds = {algo_bytes/1e3:.0f} kB")
        lines.append(f"traffic (FETCH*2 + WRITE) = {traffic:.0f} B per launch = {traffic/algo_bytes:.2f} x algorithmic; raw (FETCH + WRITE) = {raw/algo_bytes:.2f} x")
        out.update(traffic_bytes_per_launch=int(traffic), raw_bytes_per_launch=int(raw), fetch_kib=summary["FETCH_SIZE"], write_kib=summary["WRITE_SIZE"])
    with open(os.path.join(OUT, f"pmc_{tag}_hbm_traffic{suffix}.txt"), "w") as f:
        f.write("\n".join(lines) + "\n")
    with open(os.path.join(OUT, f"pmc_{tag}_hbm_traffic{suffix}.json"), "w") as f:
        json.dump(out, f, indent=1)
    print("\n".join(lines))


SQ_COUNTERS = ["SQ_WAVE_CYCLES", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_LDS", "SQ_ACTIVE_INST_SCA", "SQ_WAIT_ANY",
               "SQ_WAIT_INST_ANY", "SQ_INSTS_VALU"]


def sq_mix(tag, workload=None):
    """Where the waves of the step kernel spend their cycles (the kernel is issue-bound, not HBM-bound): one SQ pass, 8 counters.
    WAIT_ANY + WAIT_INST_ANY + ACTIVE_INST_ANY ~ WAVE_CYCLES (MI355X_MICROARCH.md, PMC slot table).  Means over the last 12 full-grid launches of the
    step kernel (= the timed region of the command; the pre-roll, warm-up, reset-time forward and compacted side launches are left out).  The JSON twin
    is what bench.py's roofline.valu quotes (SQ_INSTS_VALU per launch / the live kernel duration against the 1228.8 G wave64-instructions/s issue peak)."""
    w = WORKLOADS[workload or "fetch"]
    kernel = w["kernel"]
    suffix = f"_{workload}" if workload else ""
    extra = (["--workload", workload] if workload else []) + os.environ.get("GRX_COLLECT_EXTRA", "").split()
    d = os.path.join(OUT, f"pmc_{tag}_SQ{suffix}")
    cmd = ["rocprofv3", "--pmc"] + SQ_COUNTERS + ["--kernel-trace", "--output-format", "csv", "-d", d, "-o", "pmc", "--", sys.executable,
           os.path.join(ROOT, "bench.py"), "--steps", "12", "--warmup", "3", "--no-cpu-baseline"] + extra
    run(cmd, os.path.join(OUT, f"pmc_{tag}_SQ{suffix}.log"))
    acc = {}
    full_grid = ((w["worlds"] + 7) // 8) * 8 * 64
    for path in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        with open(path) as f:
            for row in csv.DictReader(f):
                if kernel in row.get("Kernel_Name", "") and _whole_batch(int(float(row.get("Grid_Size") or 0)), full_grid):
                    acc.setdefault(row["Counter_Name"], []).append(float(row["Counter_Value"]))
    lines = [f"rocprofv3 --pmc {' '.join(SQ_COUNTERS)} --kernel-trace --output-format csv -- python bench.py --steps 12 --warmup 3 --no-cpu-baseline {' '.join(extra)}  (MI355X, build '{tag}')",
             f"means over the last 12 full-grid STEP launches of {kernel} (the timed region; masked full-grid forward passes of resets are recognised by their wave cycles and left out):"]
    # a reset-time forward pass launched over the full grid with a mask (the kitchen's device-resident reset mask) carries the step kernel's name and grid but a tiny share
    # of its work: averaged in, it halved every count (round 5 found the kitchen's and -- before its resets were compacted -- the hammer's VALU fraction reported at half)
    wave = acc.get("SQ_WAVE_CYCLES", [])[-24:]
    steps = [i for i, x in enumerate(wave) if x >= 0.5 * max(wave)][-12:] if wave else []
    mean = {k: sum(v[-24:][i] for i in steps) / len(steps) for k, v in acc.items() if v and steps and len(v[-24:]) == len(wave)}
    for k in SQ_COUNTERS:
        lines.append(f"  {k:22s} {mean[k]:16.0f}" if k in mean else f"  {k:22s} (not collected)")
    wc = mean.get("SQ_WAVE_CYCLES")
    out = dict(mean, kernel=kernel, build=tag, build_id=_build_id(), git=_git_head(), worlds=w["worlds"])
    if wc:
        for k in ("SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_LDS", "SQ_ACTIVE_INST_SCA", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY"):
            if k in mean:
                lines.append(f"  {k} / SQ_WAVE_CYCLES = {mean[k] / wc:.3f}")
                out[k + "_frac"] = mean[k] / wc
    with open(os.path.join(OUT, f"pmc_{tag}_sq_mix{suffix}.txt"), "w") as f:
        f.write("\n".join(lines) + "\n")
    with open(os.path.join(OUT, f"pmc_{tag}_sq_mix{suffix}.json"), "w") as f:
        json.dump(out, f, indent=1)
    print("\n".join(lines))


if __name__ == "__main__":
    tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
    os.makedirs(OUT, exist_ok=True)
    if len(sys.argv) > 2 and sys.argv[2] == "sq":
        for w in (sys.argv[3:] or ["fetch"]):
            sq_mix(tag, None if w == "fetch" else w)
        sys.exit(0)
    if len(sys.argv) > 2 and sys.argv[2] == "stats":          # kernel stats only: fetch (name it "fetch") and / or other workloads
        for w in sys.argv[3:]:
            kernel_stats(tag, None if w == "fetch" else w)
        sys.exit(0)
    if len(sys.argv) > 2 and sys.argv[2] == "pmc":            # HBM traffic only (A/B of builds: GRX_HIP_LIB selects the library)
        for w in sys.argv[3:]:
            pmc(tag, None if w == "fetch" else w)
        sys.exit(0)
    if len(sys.argv) > 2 and sys.argv[2] == "workloads":      # kernel stats + HBM traffic of the other BASELINE configs
        for w in (sys.argv[3:] or ["antmaze", "hand_touch", "adroit", "hand_reach"]):
            kernel_stats(tag, w)
            pmc(tag, w)
        sys.exit(0)
    kernel_stats(tag)
    pmc(tag)
    sq_mix(tag)
