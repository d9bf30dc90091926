"""Measure the FREE-RUNNING HIP-vs-oracle error of every rollout fixture family at horizons 1 / 2 / 5 / 10 steps (tests/tolerance_cases.py::horizon_errors) on the GPU
and write gpurun_out/horizon_table.json (copied into tests/golden/tolerance_table.json under "horizons"; tests/test_gpu_horizons.py asserts against it).

    python tools/measure_horizons.py
"""
import json
import os
import sys

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from tolerance_cases import HORIZONS, measure_episodes, measure_horizons  # noqa: E402

if __name__ == "__main__":
    table = measure_horizons(sys.argv[1:] or None)
    episodes = measure_episodes(sys.argv[1:] or None)
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", "horizon_table.json"), "w") as f:
        json.dump(table, f, indent=1)
    # merged into the committed table under "horizons" (tests/test_gpu_horizons.py reads it there); the merged file also travels back through gpurun_out/
    from tolerance_cases import TABLE  # noqa: E402
    with open(TABLE) as f:
        full = json.load(f)
    full.setdefault("horizons", {}).update(table)
    full.setdefault("episodes", {}).update(episodes)
    for path in (TABLE, os.path.join(ROOT, "gpurun_out", "tolerance_table.json")):
        with open(path, "w") as f:
            json.dump(full, f, indent=1)
    for fam, row in table.items():
        for h in HORIZONS:
            r = row[str(h)]
            for comp, q in r.items():
                if isinstance(q, dict):
                    print(f"{fam:18s} h={h:2d} {comp:26s} starts {r['n_starts']:4d} posed {r['n_posed']:4d}  p50 {q['p50']:.2e} p90 {q['p90']:.2e} max {q['max']:.2e} within1e-4 {100 * q['frac_within_1e-4']:5.1f}% | posed: "
                          f"max {q['max_posed'] if q['max_posed'] is None else format(q['max_posed'], '.2e')} within {q['frac_within_1e-4_posed'] if q['frac_within_1e-4_posed'] is None else format(100 * q['frac_within_1e-4_posed'], '5.1f')}%")
    print("whole fixture episodes, free-running from their first state (one world per episode):")
    for fam, row in episodes.items():
        for comp, q in row.items():
            if isinstance(q, dict):
                print(f"{fam:18s} {comp:26s} {row['episodes']:2d} episodes of {min(row['steps'])}-{max(row['steps'])} steps: within 1e-4 throughout {q['episodes_within_1e-4_throughout']:2d}; error at the last step median {q['final_median']:.2e} max {q['final_max']:.2e}; "
                      f"worst step median {q['worst_median']:.2e} max {q['worst_max']:.2e}; steps before the first 1e-4: median {q['steps_before_first_1e-4_median']:.0f} min {q['steps_before_first_1e-4_min']}")
