"""Compile the reference's MJCF assets into packaged model blobs (gymnasium_robotics_amd/models/*.npz).

Run in a container where the reference tree is mounted (it is not present on the GPU box):
    python tools/compile_models.py [/root/reference/gymnasium_robotics/envs/assets] [--only fetch,hand,maze,adroit,kitchen]
The blobs contain only numeric tables derived from the MJCF/STL inputs (SURVEY.md §2 row 10:
"read-only input -- the new framework must parse these (or a pre-compiled blob derived from them)").
"""
import argparse
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from gymnasium_robotics_amd.mjcf import compile_mjcf, save_model  # noqa: E402

OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "gymnasium_robotics_amd", "models")
DIMS = ("nq", "nv", "nu", "nbody", "ngeom", "nsite", "npair")


def _save(m, name, what, **extra):
    out = os.path.join(OUT, name)
    save_model(m, out)
    print(what, "->", out, {k: m.dim(k) for k in DIMS}, "origin:", m.origin.tolist(), "unsupported pairs:", m.info["unsupported_pairs"], extra, f"{os.path.getsize(out) / 1024:.0f} KiB", flush=True)


def fetch(assets):
    from gymnasium_robotics_amd.envs.fetch import FETCH_CAPACITY

    for xml in ("fetch/reach.xml", "fetch/push.xml", "fetch/slide.xml", "fetch/pick_and_place.xml"):
        # joint-box gates of the hull pairs (mjcf/pair_gates.py): minutes per model, packaged blobs only
        m = compile_mjcf(os.path.join(assets, xml), capacity=dict(FETCH_CAPACITY, pair_gates=True))
        _save(m, os.path.splitext(os.path.basename(xml))[0] + ".npz", xml)


def hand(assets):
    from gymnasium_robotics_amd.envs.hand import HAND_MANIP_CAPACITY, HAND_ORIGIN, HAND_REACH_COMPILE
    from gymnasium_robotics_amd.envs.manipulate_spec import drop_target_body, touch_filter

    m = compile_mjcf(os.path.join(assets, "hand", "reach.xml"), origin=HAND_ORIGIN, **HAND_REACH_COMPILE)
    _save(m, "hand_reach.npz", "hand/reach.xml", tendons=len(m.tables["tendon_adr"]))
    for xml, name, tf in (("manipulate_block.xml", "hand_block.npz", None), ("manipulate_block_touch_sensors.xml", "hand_block_touch.npz", touch_filter),
                          ("manipulate_pen.xml", "hand_pen.npz", None), ("manipulate_pen_touch_sensors.xml", "hand_pen_touch.npz", touch_filter),
                          ("manipulate_egg.xml", "hand_egg.npz", None), ("manipulate_egg_touch_sensors.xml", "hand_egg_touch.npz", touch_filter)):
        # (no target body: manipulate_spec.drop_target_body; touch keeps contact data out of the LDS overlay: a slightly smaller pool)
        m = compile_mjcf(os.path.join(assets, "hand", xml), mutate=drop_target_body, touch_filter=tf, keep_sites=[], origin=HAND_ORIGIN,
                         capacity=dict(HAND_MANIP_CAPACITY, jpool=928) if tf else HAND_MANIP_CAPACITY)
        _save(m, name, f"hand/{xml} (no target body)", touch_zones=len(m.tables["touch_body"]))


def maze(assets):
    from gymnasium_robotics_amd.envs.maze_spec import ANT_MAZE_HEIGHT, ANT_MAZE_SIZE_SCALING, MAPS, POINT_MAZE_HEIGHT, POINT_MAZE_SIZE_SCALING, Maze
    from gymnasium_robotics_amd.envs.point_maze import ANT_CAPACITY

    for layout in ("UMaze", "Open", "Medium", "Large"):
        mz = Maze(MAPS[layout], POINT_MAZE_SIZE_SCALING, POINT_MAZE_HEIGHT)
        _save(compile_mjcf(os.path.join(assets, "point", "point.xml"), mutate=mz.add_walls), f"point_{layout}.npz", f"point.xml + {layout} walls")
    for layout in ("UMaze", "Open", "Medium", "Large"):
        mz = Maze(MAPS[layout], ANT_MAZE_SIZE_SCALING, ANT_MAZE_HEIGHT)
        _save(compile_mjcf(os.path.join(assets, "..", "mujoco", "assets", "ant.xml"), mutate=mz.add_walls, capacity=ANT_CAPACITY), f"ant_{layout}.npz", f"ant.xml + {layout} walls")


def adroit(assets):
    from gymnasium_robotics_amd.envs.adroit_spec import SPECS, apply_actuator_overrides

    for task, spec in SPECS.items():
        m = apply_actuator_overrides(compile_mjcf(os.path.join(assets, "adroit_hand", spec["xml"]), **spec["compile"]))   # the constructors' gain / bias rewrite is baked in
        _save(m, spec["npz"], f"adroit_hand/{spec['xml']}")


def kitchen(assets):
    from gymnasium_robotics_amd.envs.kitchen_spec import KITCHEN_CAPACITY, load_kitchen_model

    # compiles kitchen_env_model.xml (joint-box gates of the arm's hull pairs: mjcf/pair_gates.py) and attaches the numbers of franka_config.xml (model.info["franka_config"])
    m = load_kitchen_model(assets, capacity=dict(KITCHEN_CAPACITY, pair_gates=True))
    _save(m, "kitchen.npz", "kitchen_franka/kitchen_assets/kitchen_env_model.xml", joint_equalities=len(m.tables["jeq_eq"]))


FAMILIES = dict(fetch=fetch, hand=hand, maze=maze, adroit=adroit, kitchen=kitchen)

if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("assets", nargs="?", default="/root/reference/gymnasium_robotics/envs/assets")
    ap.add_argument("--only", default=None, help="comma-separated families: " + ",".join(FAMILIES))
    a = ap.parse_args()
    os.makedirs(OUT, exist_ok=True)
    for fam in (a.only.split(",") if a.only else FAMILIES):
        FAMILIES[fam](a.assets)
