"""Measure the teacher-forced HIP-vs-oracle error quantiles of every served family on the GPU and write tests/golden/tolerance_table.json
(the table tests/test_gpu_tolerance_table.py asserts against, and DESIGN.md quotes).  Run on the GPU box:

    python tools/measure_tolerances.py            # prints the table and writes gpurun_out/tolerance_table.json (copy it to tests/golden/)
"""
import json
import os
import sys

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from tolerance_cases import measure_all  # noqa: E402

if __name__ == "__main__":
    table = measure_all()
    from tolerance_cases import TABLE  # noqa: E402
    if os.path.exists(TABLE):      # the free-running section (tools/measure_horizons.py) lives in the same file
        with open(TABLE) as f:
            old = json.load(f)
        for sec in ("horizons", "episodes", "reference_sensitivity", "reference_sensitivity_horizons"):      # written by tools/measure_horizons.py / tools/emu_tolerances.py --sensitivity --json
            if sec in old:
                table[sec] = old[sec]
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", "tolerance_table.json"), "w") as f:
        json.dump(table, f, indent=1)
    for fam, row in table.items():
        if fam in ("horizons", "episodes", "reference_sensitivity", "reference_sensitivity_horizons"):
            continue
        for comp, q in row.items():
            if isinstance(q, dict):
                print(f"{fam:18s} {comp:18s} p50 {q['p50']:.2e}  p90 {q['p90']:.2e}  p99 {q['p99']:.2e}  max {q['max']:.2e}  within 1e-4: {100 * q['frac_within_1e-4']:.1f} %"
                      f"  max away from activation boundaries {q['max_away_from_boundary'] if q['max_away_from_boundary'] is None else format(q['max_away_from_boundary'], '.2e')}"
                      f"  (n = {row['n']}, away: {row['n_away_from_activation_boundary']})" + (f"  outliers (snapshot, error): {q['outliers']}" if q["outliers"] else ""))
