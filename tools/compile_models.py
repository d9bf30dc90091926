"""Compile the reference's MJCF assets into packaged model blobs (gymnasium_robotics_amd/models/*.npz).

Run in a container where the reference tree is mounted (it is not present on the GPU box):
    python tools/compile_models.py [/root/reference/gymnasium_robotics/envs/assets]
The blobs contain only numeric tables derived from the MJCF/STL inputs (SURVEY.md §2 row 10:
"read-only input -- the new framework must parse these (or a pre-compiled blob derived from them)").
"""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from gymnasium_robotics_amd.mjcf import compile_mjcf, save_model  # noqa: E402

ASSETS = sys.argv[1] if len(sys.argv) > 1 else "/root/reference/gymnasium_robotics/envs/assets"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "gymnasium_robotics_amd", "models")
os.makedirs(OUT, exist_ok=True)
from gymnasium_robotics_amd.envs.fetch import FETCH_CAPACITY  # noqa: E402

for xml in ("fetch/reach.xml", "fetch/push.xml", "fetch/slide.xml", "fetch/pick_and_place.xml"):
    m = compile_mjcf(os.path.join(ASSETS, xml), capacity=dict(FETCH_CAPACITY, pair_gates=True))      # joint-box gates of the hull pairs (mjcf/pair_gates.py): minutes per model, packaged blobs only
    # keep hull vertices only for meshes that take part in a supported pair
    out = os.path.join(OUT, os.path.splitext(os.path.basename(xml))[0] + ".npz")
    save_model(m, out)
    print(xml, "->", out, {k: m.dim(k) for k in ("nq", "nv", "nu", "nbody", "ngeom", "npair")}, "unsupported pairs:", m.info["unsupported_pairs"],
          f"{os.path.getsize(out) / 1024:.0f} KiB")

from gymnasium_robotics_amd.envs.hand import HAND_REACH_COMPILE  # noqa: E402

m = compile_mjcf(os.path.join(ASSETS, "hand", "reach.xml"), **HAND_REACH_COMPILE)
out = os.path.join(OUT, "hand_reach.npz")
save_model(m, out)
print("hand/reach.xml ->", out, {k: m.dim(k) for k in ("nq", "nv", "nu", "nbody", "ngeom", "npair")}, "tendons:", len(m.tables["tendon_adr"]),
      "unsupported pairs:", m.info["unsupported_pairs"], f"{os.path.getsize(out) / 1024:.0f} KiB")

from gymnasium_robotics_amd.envs.hand import HAND_MANIP_CAPACITY  # noqa: E402
from gymnasium_robotics_amd.envs.manipulate_spec import drop_target_body, touch_filter  # noqa: E402

m = compile_mjcf(os.path.join(ASSETS, "hand", "manipulate_block_touch_sensors.xml"), mutate=drop_target_body, touch_filter=touch_filter, keep_sites=[],
                 capacity=dict(HAND_MANIP_CAPACITY, jpool=928))
out = os.path.join(OUT, "hand_block_touch.npz")
save_model(m, out)
print("hand/manipulate_block_touch_sensors.xml (no target body) ->", out, {k: m.dim(k) for k in ("nq", "nv", "nu", "nbody", "ngeom", "npair")},
      "touch zones:", len(m.tables["touch_body"]), f"{os.path.getsize(out) / 1024:.0f} KiB")

m = compile_mjcf(os.path.join(ASSETS, "hand", "manipulate_block.xml"), mutate=drop_target_body, keep_sites=[], capacity=HAND_MANIP_CAPACITY)
out = os.path.join(OUT, "hand_block.npz")
save_model(m, out)
print("hand/manipulate_block.xml (no target body) ->", out, {k: m.dim(k) for k in ("nq", "nv", "nu", "nbody", "ngeom", "npair")},
      "unsupported pairs:", m.info["unsupported_pairs"], f"{os.path.getsize(out) / 1024:.0f} KiB")

for xml, name, tf in (("manipulate_pen.xml", "hand_pen.npz", None), ("manipulate_pen_touch_sensors.xml", "hand_pen_touch.npz", touch_filter),
                      ("manipulate_egg.xml", "hand_egg.npz", None), ("manipulate_egg_touch_sensors.xml", "hand_egg_touch.npz", touch_filter)):
    m = compile_mjcf(os.path.join(ASSETS, "hand", xml), mutate=drop_target_body, touch_filter=tf, keep_sites=[],
                     capacity=dict(HAND_MANIP_CAPACITY, jpool=928) if tf else HAND_MANIP_CAPACITY)
    out = os.path.join(OUT, name)
    save_model(m, out)
    print(f"hand/{xml} (no target body) ->", out, {k: m.dim(k) for k in ("nq", "nv", "nu", "nbody", "ngeom", "npair")}, "touch zones:", len(m.tables["touch_body"]),
          f"{os.path.getsize(out) / 1024:.0f} KiB")

from gymnasium_robotics_amd.envs.maze_spec import MAPS, POINT_MAZE_HEIGHT, POINT_MAZE_SIZE_SCALING, Maze  # noqa: E402

for layout in ("UMaze", "Open", "Medium", "Large"):
    maze = Maze(MAPS[layout], POINT_MAZE_SIZE_SCALING, POINT_MAZE_HEIGHT)
    m = compile_mjcf(os.path.join(ASSETS, "point", "point.xml"), mutate=maze.add_walls)
    out = os.path.join(OUT, f"point_{layout}.npz")
    save_model(m, out)
    print("point.xml +", layout, "walls ->", out, {k: m.dim(k) for k in ("nq", "nv", "nu", "ngeom", "npair")}, f"{os.path.getsize(out) / 1024:.0f} KiB")

from gymnasium_robotics_amd.envs.maze_spec import ANT_MAZE_HEIGHT, ANT_MAZE_SIZE_SCALING  # noqa: E402
from gymnasium_robotics_amd.envs.point_maze import ANT_CAPACITY  # noqa: E402

for layout in ("UMaze", "Open", "Medium", "Large"):
    maze = Maze(MAPS[layout], ANT_MAZE_SIZE_SCALING, ANT_MAZE_HEIGHT)
    m = compile_mjcf(os.path.join(ASSETS, "..", "mujoco", "assets", "ant.xml"), mutate=maze.add_walls, capacity=ANT_CAPACITY)
    out = os.path.join(OUT, f"ant_{layout}.npz")
    save_model(m, out)
    print("ant.xml +", layout, "walls ->", out, {k: m.dim(k) for k in ("nq", "nv", "nu", "ngeom", "npair")}, f"{os.path.getsize(out) / 1024:.0f} KiB")

from gymnasium_robotics_amd.envs.adroit_spec import SPECS as ADROIT_SPECS, apply_actuator_overrides  # noqa: E402

for task, spec in ADROIT_SPECS.items():
    m = apply_actuator_overrides(compile_mjcf(os.path.join(ASSETS, "adroit_hand", spec["xml"]), **spec["compile"]))   # the constructors' gain / bias rewrite is baked in
    out = os.path.join(OUT, spec["npz"])
    save_model(m, out)
    print(f"adroit_hand/{spec['xml']} ->", out, {k: m.dim(k) for k in ("nq", "nv", "nu", "nbody", "ngeom", "nsite", "npair")}, "unsupported pairs:", m.info["unsupported_pairs"],
          f"{os.path.getsize(out) / 1024:.0f} KiB")

from gymnasium_robotics_amd.envs.kitchen_spec import load_kitchen_model  # noqa: E402

from gymnasium_robotics_amd.envs.kitchen_spec import KITCHEN_CAPACITY  # noqa: E402

m = load_kitchen_model(ASSETS, capacity=dict(KITCHEN_CAPACITY, pair_gates=True))      # (joint-box gates of the arm's hull pairs: mjcf/pair_gates.py) compiles kitchen_env_model.xml and attaches the numbers of franka_config.xml (model.info["franka_config"])
out = os.path.join(OUT, "kitchen.npz")
save_model(m, out)
print("kitchen_franka/kitchen_assets/kitchen_env_model.xml ->", out, {k: m.dim(k) for k in ("nq", "nv", "nu", "nbody", "ngeom", "npair")}, "joint equalities:", len(m.tables["jeq_eq"]),
      "unsupported pairs:", m.info["unsupported_pairs"], f"{os.path.getsize(out) / 1024:.0f} KiB")
