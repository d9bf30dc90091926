"""bench.py quotes PMC counters (roofline.traffic / roofline.valu) from committed summaries: only those measured on the device code that is loaded."""
import json
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bench  # noqa: E402


def _write(root, name, payload):
    os.makedirs(os.path.join(root, "profiles"), exist_ok=True)
    with open(os.path.join(root, "profiles", name), "w") as f:
        json.dump(payload, f)


def test_a_summary_of_another_build_is_refused(tmp_path):
    tag = bench.PROFILE_TAGS[0]
    _write(tmp_path, f"pmc_{tag}_hbm_traffic.json", {"build_id": "aaaa", "traffic_bytes_per_launch": 123})
    _write(tmp_path, f"pmc_{tag}_sq_mix_kitchen.json", {"build_id": "aaaa", "SQ_INSTS_VALU": 7})
    d, src = bench.pmc_summary("hbm_traffic", "fetch", 4096, 4096, "aaaa", root=str(tmp_path))
    assert d["traffic_bytes_per_launch"] == 123 and src.endswith(f"pmc_{tag}_hbm_traffic.json")
    d, src = bench.pmc_summary("hbm_traffic", "fetch", 4096, 4096, "bbbb", root=str(tmp_path))       # the library changed since the counters were collected
    assert d is None and src.startswith("STALE") and "aaaa" in src and "bbbb" in src
    d, src = bench.pmc_summary("hbm_traffic", "fetch", 8192, 4096, "aaaa", root=str(tmp_path))       # another batch size than the one profiled
    assert d is None and "8192" in src
    d, src = bench.pmc_summary("sq_mix", "kitchen", 16384, 16384, "aaaa", root=str(tmp_path))
    assert d["SQ_INSTS_VALU"] == 7
    assert bench.pmc_summary("sq_mix", "antmaze", 8192, 8192, "aaaa", root=str(tmp_path)) == (None, None)      # nothing collected: nothing quoted


def test_an_unstamped_summary_is_refused(tmp_path):
    tag = bench.PROFILE_TAGS[0]
    _write(tmp_path, f"pmc_{tag}_hbm_traffic.json", {"traffic_bytes_per_launch": 1})       # a file of an earlier round: no build_id
    d, src = bench.pmc_summary("hbm_traffic", "fetch", 4096, 4096, "cccc", root=str(tmp_path))
    assert d is None and "unstamped" in src


def test_the_committed_summaries_carry_a_build_id():
    root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "profiles")
    tag = bench.PROFILE_TAGS[0]
    names = [n for n in os.listdir(root) if n.startswith(f"pmc_{tag}_") and n.endswith(".json")]
    assert names, "no PMC summary of the current round is committed"
    for n in names:
        with open(os.path.join(root, n)) as f:
            assert json.load(f).get("build_id"), n


def test_a_summary_without_its_figure_is_refused_not_fatal(tmp_path):
    """a counter pass that matched no launch of the step kernel (round 6: the split step's grid is 2 x the worlds, the collector filtered it out) writes a summary without
    `traffic_bytes_per_launch`: the bench line must say so and carry `traffic: null`, not die on a KeyError"""
    tag = bench.PROFILE_TAGS[0]
    _write(tmp_path, f"pmc_{tag}_hbm_traffic.json", {"build_id": "aaaa", "algorithmic_bytes_per_launch": 1})
    _write(tmp_path, f"pmc_{tag}_sq_mix.json", {"build_id": "aaaa", "kernel": "k"})
    for kind in ("hbm_traffic", "sq_mix"):
        d, src = bench.pmc_summary(kind, "fetch", 4096, 4096, "aaaa", root=str(tmp_path))
        assert d is None and src.startswith("INCOMPLETE")


def test_the_collector_counts_split_step_launches_as_whole_batch_launches():
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tools"))
    import collect_profiles as cp

    full = 4096 * 64
    assert cp._whole_batch(full, full) and cp._whole_batch(2 * full, full) and cp._whole_batch(8 * full, full)
    assert not cp._whole_batch(82 * 64, full) and not cp._whole_batch(9 * full, full) and not cp._whole_batch(full + 64, full) and not cp._whole_batch(0, full)
