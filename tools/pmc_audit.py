"""What saturates when the Fetch step kernel runs: several separate rocprofv3 --pmc passes (counters only, --kernel-trace, no other trace domain) of the
same short bench command, each with one group of counters; means over the full-grid launches of the step kernel.  Run ON the GPU box:

    python tools/pmc_audit.py [workload] > gpurun_out/pmc_audit_<tag>.txt

Groups: instruction mix (what a wave issues), instruction fetch / instruction cache (the kernel's code is ~250 KB, the I-cache 64 KB per two CUs), LDS (bank conflicts),
scalar data cache, vector L1 / L2 (model tables and scratch).  A counter the device does not offer is reported as such and skipped."""
import csv
import glob
import os
import subprocess
import sys

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
OUT = os.path.join(ROOT, "gpurun_out")
sys.path.insert(0, ROOT)
from bench import WORKLOADS  # noqa: E402

GROUPS = [
    ("instruction mix", ["SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES", "SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_INSTS_SMEM", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR"]),
    ("issue / wait", ["SQ_WAVE_CYCLES", "SQ_ACTIVE_INST_ANY", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_WAIT_INST_LDS", "SQ_INST_CYCLES_VMEM_RD", "SQ_INST_CYCLES_SMEM", "SQ_INSTS_BRANCH"]),
    ("latency (LEVEL = in-flight instructions accumulated per cycle: LEVEL / INSTS = mean latency in the counter's cycle unit)", ["SQ_INST_LEVEL_VMEM", "SQ_INST_LEVEL_SMEM", "SQ_INST_LEVEL_LDS", "SQ_INSTS_VMEM_RD", "SQ_INSTS_SMEM", "SQ_INSTS_LDS", "SQ_WAVE_CYCLES"]),
    ("instruction fetch", ["SQ_WAVE_CYCLES", "SQ_IFETCH", "SQ_IFETCH_LEVEL", "SQC_ICACHE_REQ", "SQC_ICACHE_HITS", "SQC_ICACHE_MISSES", "SQC_ICACHE_MISSES_DUPLICATE"]),
    ("scalar data cache", ["SQC_DCACHE_REQ", "SQC_DCACHE_HITS", "SQC_DCACHE_MISSES", "SQC_DCACHE_MISSES_DUPLICATE", "SQC_TC_REQ", "SQC_TC_INST_REQ", "SQC_TC_DATA_READ_REQ"]),
    ("LDS", ["SQ_LDS_BANK_CONFLICT", "SQ_LDS_IDX_ACTIVE", "SQ_LDS_ADDR_CONFLICT", "SQ_LDS_UNALIGNED_STALL", "SQ_LDS_MEM_VIOLATIONS", "SQ_ACTIVE_INST_LDS", "SQ_INSTS_LDS"]),
    ("vector L1 a", ["TCP_TOTAL_CACHE_ACCESSES_sum", "TCP_TCC_READ_REQ_sum"]),      # (derived sums over 256 TCPs: two per pass, a larger request exceeds what one pass can collect)
    ("vector L1 b", ["TCP_TCC_WRITE_REQ_sum", "TCP_PENDING_STALL_CYCLES_sum"]),
    ("L2", ["TCC_REQ_sum", "TCC_HIT_sum", "TCC_MISS_sum"]),
]


def main():
    workload = sys.argv[1] if len(sys.argv) > 1 else "fetch"
    w = WORKLOADS[workload]
    extra = os.environ.get("GRX_AUDIT_EXTRA", "").split()
    full_grid = ((w["worlds"] + 7) // 8) * 8 * 64
    print(f"PMC audit of {w['kernel']} ({w['env_id']}, {w['worlds']} worlds): python bench.py --workload {workload} --steps 12 --warmup 3 --preroll 60 --no-cpu-baseline {' '.join(extra)}")
    print("means per full-grid launch of the step kernel (last 12 = the timed region)", flush=True)
    os.makedirs(OUT, exist_ok=True)
    only = os.environ.get("GRX_AUDIT_GROUPS")      # e.g. "2,3": just those groups
    for gi, (name, counters) in enumerate(GROUPS):
        if only and str(gi) not in only.split(","):
            continue
        d = os.path.join(OUT, f"pmc_audit_{gi}")
        cmd = ["rocprofv3", "--pmc"] + counters + ["--kernel-trace", "--output-format", "csv", "-d", d, "-o", "pmc", "--", sys.executable, os.path.join(ROOT, "bench.py"),
               "--workload", workload, "--steps", "12", "--warmup", "3", "--preroll", "60", "--no-cpu-baseline"] + extra
        with open(os.path.join(OUT, f"pmc_audit_{gi}.log"), "w") as f:
            try:      # (a request the hardware cannot collect aborts rocprofv3 and leaves it hanging: bounded per pass)
                subprocess.run(cmd, cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp"), stdout=f, stderr=subprocess.STDOUT, check=False, timeout=150)
            except subprocess.TimeoutExpired:
                print(f"-- {name}: pass timed out (see gpurun_out/pmc_audit_{gi}.log)", flush=True)
                continue
        acc = {}
        for path in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            with open(path) as f:
                for row in csv.DictReader(f):
                    if w["kernel"] in row.get("Kernel_Name", "") and int(float(row.get("Grid_Size") or 0)) == full_grid:
                        acc.setdefault(row["Counter_Name"], []).append(float(row["Counter_Value"]))
        print(f"-- {name}", flush=True)
        for c in counters:
            v = acc.get(c, [])[-12:]
            print(f"   {c:32s} {sum(v) / len(v):18.0f}   ({len(v)} launches)" if v else f"   {c:32s} (not collected: see gpurun_out/pmc_audit_{gi}.log)", flush=True)
        subprocess.run(["rm", "-rf", d])


if __name__ == "__main__":
    main()
