"""The id registry (gymnasium_robotics_amd.make_vec / registered_env_ids) against the reference's registration table
(gymnasium_robotics/__init__.py:12-1201).  Host logic only -- no torch, no device."""
import pytest

import gymnasium_robotics_amd as grx
from gymnasium_robotics_amd.envs import fetch_spec, hand_spec, manipulate_spec, maze_spec


def test_registry_counts_and_uniqueness():
    ids = grx.registered_env_ids()
    assert len(ids) == len(set(ids))
    fam = {}
    for i in ids:
        fam.setdefault(grx.env_family(i), []).append(i)
    # 4 Fetch tasks x {sparse, Dense}; HandReach x 2; 11 block/egg/pen bases, 8 of them with two touch twins, x 2; 10 maps x 2 x 2 agents
    assert {k: len(v) for k, v in fam.items()} == {"fetch": 8, "hand_reach": 2, "hand_manipulate": (8 * 3 + 3) * 2, "point_maze": 20, "ant_maze": 20, "adroit": 16, "kitchen": 1}


def test_every_registered_id_parses():
    for i in grx.registered_env_ids():
        f = grx.env_family(i)
        if f == "fetch":
            base, rt = fetch_spec.parse_env_id(i)
            assert base in fetch_spec.FETCH_TASKS
        elif f == "hand_reach":
            rt = hand_spec.parse_hand_reach_id(i)
            rt = rt[-1] if isinstance(rt, tuple) else rt
        elif f == "hand_manipulate":
            tp, tr, rt, touch = manipulate_spec.parse_block_id(i)
            assert manipulate_spec.object_of(i) in manipulate_spec.OBJECTS
            assert (touch != "off") == ("TouchSensors" in i)
        elif f == "kitchen":
            assert i == "FrankaKitchen-v1"
            continue
        elif f == "adroit":
            from gymnasium_robotics_amd.envs import adroit_spec

            task, rt = adroit_spec.parse_adroit_id(i)
            assert task in adroit_spec.SPECS and task in i.lower() and rt == ("sparse" if "Sparse" in i else "dense")
            continue   # the Adroit ids mark the SPARSE variant in the name (dense is the default: __init__.py:1082-1094)
        elif f == "point_maze":
            name, rt, limit = maze_spec.parse_point_maze_id(i)
            assert limit == maze_spec.POINT_MAX_EPISODE_STEPS[name.split("_")[0]]
        else:
            name, rt, limit = maze_spec.parse_ant_maze_id(i)
            assert limit == maze_spec.ANT_MAX_EPISODE_STEPS[name.split("_")[0]]
        assert rt == ("dense" if "Dense-v" in i else "sparse"), i


def test_every_reference_id_is_served():
    """gymnasium_robotics/__init__.py registers 8 Fetch + 2 HandReach + 54 HandManipulate + 40 maze + 8 Adroit (v2; the -v1 aliases are extra here) +
    FrankaKitchen-v1: nothing is left for UnsupportedEnvError (kept for ids a later reference version may add)."""
    assert len(grx.registered_env_ids()) == 8 + 2 + 54 + 40 + 16 + 1 and not grx._NOT_SERVED


def test_unknown_id():
    with pytest.raises(KeyError, match="unknown env id"):
        grx.make_vec("FetchReach-v9")
