"""VERDICT r04 item 5 ("two worlds per wavefront for nv <= 32 ... or commit the A/B that kills it"): what would halving the INSTRUCTIONS per world buy?  Measured, not argued:
the step kernel of AntMaze (14 dofs, no hull pairs, 168 VGPRs) and of FetchPickAndPlace is run with fewer worlds resident per CU (extra dynamic LDS per workgroup,
GRX_LDS_PAD_BYTES: no code change, same worlds, same results) and the throughput is recorded against the number of resident worlds.

  * If a wave's time through its step were set by INSTRUCTION ISSUE (waves compete for the SIMD), removing resident waves would speed the remaining ones up and the
    throughput would fall slower than the occupancy: a wave that did two worlds' work in fewer instructions would then pay off.
  * If it is set by LATENCY (dependent chains through LDS round trips, readlane eliminations, s_waitcnt), a wave takes the same time whoever else is resident: throughput is
    proportional to the resident worlds, and LDS -- not wave slots, not issue slots -- decides how many are resident.  Two worlds in one wave then need the SAME LDS, run the
    SAME dependent chains (plus the divergence of their iteration counts), and buy nothing.

    python tools/occupancy_sweep.py [antmaze fetch]      -> gpurun_out/ab_r05_two_worlds_occupancy.txt
"""
import json
import os
import subprocess
import sys

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)
GRANULE = 1280      # LDS allocation granule (bytes) of a workgroup on gfx950


def lds_bytes(workload):
    code = ("import sys; sys.path.insert(0, %r); import bench; e = bench.make_env(%r, 64, 'cuda:0', 0); print(e.lds_bytes if hasattr(e, 'lds_bytes') else e._L.grx_model_lds_bytes(e._h))" % (ROOT, workload))
    return int(subprocess.run([sys.executable, "-c", code], capture_output=True, text=True).stdout.strip().splitlines()[-1])


def main(workloads):
    out = []
    for w in workloads:
        base = lds_bytes(w)
        per = -(-base // GRANULE) * GRANULE
        natural = (160 * 1024) // per
        out.append(f"== {w}: working set {base} B per world -> {per} B allocated, {natural} worlds per CU by LDS")
        rows = []
        for k in [natural] + [x for x in (12, 10, 8, 7, 6, 5, 4, 3, 2) if x < natural]:
            want = ((160 * 1024) // k) // GRANULE * GRANULE          # largest allocation that still lets k workgroups share a CU
            pad = max(0, want - per) if k < natural else 0
            env = dict(os.environ, GRX_LDS_PAD_BYTES=str(pad))
            r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--workload", w, "--steps", "30", "--warmup", "5", "--preroll", "0" if w == "antmaze" else "-1", "--no-cpu-baseline"],
                               capture_output=True, text=True, env=env)
            line = json.loads(r.stdout.strip().splitlines()[-1])
            rows.append((k, pad, line["value"], line["roofline"]["kernel_ms"]))
            out.append(f"  resident worlds per CU <= {k:2d} (pad {pad:6d} B): {line['value']:10.0f} env-steps/s, step kernel {line['roofline']['kernel_ms']:.3f} ms")
        v0, k0 = rows[0][2], rows[0][0]
        for k, pad, v, ms in rows[1:]:
            out.append(f"    {k:2d} / {k0} of the worlds resident ({k / k0:.2f}) -> {v / v0:.2f} of the throughput; per resident world {v / v0 / (k / k0):.2f} x as fast")
    text = "\n".join(out)
    print(text)
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", "ab_r05_two_worlds_occupancy.txt"), "w") as f:
        f.write(text + "\n")


if __name__ == "__main__":
    main(sys.argv[1:] or ["antmaze", "fetch"])
