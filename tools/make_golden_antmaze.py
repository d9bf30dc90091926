"""Golden fixture for BASELINE.json configs[3] itself: teacher-forcing snapshots of AntMaze_Large_Diverse_GR-v5
(/root/reference/gymnasium_robotics/__init__.py:936-958; walls from envs/maze/maze_v4.py:179-212; step envs/maze/ant_maze_v5.py:295-320) on the fp64
oracle, with the ant pushed against the walls of its cell so that most snapshots carry wall-lattice contacts in some substep (under random actions
the ant flails in the middle of a 4 m cell and never meets a wall).

Every segment: reset(seed) -> a few random-action steps that land the ant on the floor -> the whole ant is translated (a pure translation of the
free joint is a valid state while nothing touches a wall) until its leading geom is a few millimetres from a wall face (or from two faces: a corner of
the cell) -> a velocity toward the wall -> recorded steps.  The substeps are run one at a time so that wall contacts of EVERY substep are counted.

    python tools/make_golden_antmaze.py   ->  tests/golden/ant_Large_teacher.npz
"""
import os
import sys

import numpy as np

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)
from gymnasium_robotics_amd.envs.maze_spec import ANT_MAZE_HEIGHT, ANT_MAZE_SIZE_SCALING, MAPS, Maze  # noqa: E402
from gymnasium_robotics_amd.envs.point_maze import load_point_maze_model  # noqa: E402
from oracle.maze_oracle import OracleAntMazeEnv  # noqa: E402

LAYOUT = "Large_Diverse_GR"
SEGMENTS, LAND_STEPS, REC_STEPS = 48, 12, 5


def wall_contacts(sim, wall_geoms):
    return [c for c in sim.contacts() if int(c[7]) in wall_geoms or int(c[8]) in wall_geoms]


def push_to_wall(sim, d, wall_geoms, gap):
    """translate the ant along d until the first wall contact appears (bisection on the offset), then back off by `gap`"""
    base = sim.qpos[:2].copy()

    def touching(off):
        sim.qpos[:2] = base + d * off
        sim.forward()
        return len(wall_contacts(sim, wall_geoms)) > 0

    lo, hi = 0.0, 0.0
    while not touching(hi):
        lo, hi = hi, hi + 0.25
        assert hi < 4.0
    for _ in range(30):
        mid = 0.5 * (lo + hi)
        lo, hi = (lo, mid) if touching(mid) else (mid, hi)
    sim.qpos[:2] = base + d * (lo - gap)
    sim.forward()


if __name__ == "__main__":
    maze = Maze(MAPS[LAYOUT], ANT_MAZE_SIZE_SCALING, ANT_MAZE_HEIGHT)
    model = load_point_maze_model(maze, LAYOUT, None, "ant")
    wall_geoms = set(int(g) for g in model.tables["grid_wall_geom"].ravel() if g >= 0)
    env = OracleAntMazeEnv(model, maze)
    s = env.sim
    rec = {k: [] for k in ("qpos", "qvel", "qacc_ws", "goal", "action", "obs", "achieved", "reward", "success", "qpos_next", "qvel_next", "ncon", "nefc",
                           "wall_contact_substeps", "floor_contact_substeps", "activation_gap", "segment")}
    for seg in range(SEGMENTS):
        rng = np.random.default_rng(1000 + seg)
        env.reset(seed=200 + seg)
        for t in range(LAND_STEPS):
            env.step(rng.uniform(-1, 1, 8))
        xy = s.qpos[:2].copy()
        col, row = int(np.floor((xy[0] + maze.x_map_center) / ANT_MAZE_SIZE_SCALING)), int(np.floor((maze.y_map_center - xy[1]) / ANT_MAZE_SIZE_SCALING))
        nb = [(np.array([dx, dy], float), maze.maze_map[r][c] == 1) for dx, dy, (r, c) in ((1, 0, (row, col + 1)), (-1, 0, (row, col - 1)), (0, 1, (row - 1, col)), (0, -1, (row + 1, col)))]
        walls = [d for d, w in nb if w]
        d = walls[rng.integers(len(walls))]
        perp = [e for e in walls if abs(e @ d) < 0.5]
        yaw = rng.uniform(-np.pi, np.pi)   # a rotation about z composed in front of the torso's orientation keeps the landed pose valid
        qz, q = np.array([np.cos(yaw / 2), 0, 0, np.sin(yaw / 2)]), s.qpos[3:7].copy()
        s.qpos[3:7] = [qz[0] * q[0] - qz[3] * q[3], qz[0] * q[1] - qz[3] * q[2], qz[0] * q[2] + qz[3] * q[1], qz[0] * q[3] + qz[3] * q[0]]
        s.qpos[:2] = maze.cell_rowcol_to_xy((row, col))
        c, sn = np.cos(yaw), np.sin(yaw)
        s.qvel[:2] = [c * s.qvel[0] - sn * s.qvel[1], sn * s.qvel[0] + c * s.qvel[1]]
        gap = rng.uniform(0.0, 0.02)
        push_to_wall(s, d, wall_geoms, gap)
        if perp and seg % 3 == 0:   # a corner of the cell: two wall faces at once
            push_to_wall(s, perp[rng.integers(len(perp))], wall_geoms, rng.uniform(0.0, 0.02))
            d = d + perp[0] * 0.5
        s.qvel[:2] += d * rng.uniform(0.3, 1.5)
        drive = rng.uniform(-1, 1, 8)
        for t in range(REC_STEPS):
            a = np.clip(drive + 0.5 * rng.uniform(-1, 1, 8), -1, 1).astype(np.float32)
            if t in (2, 4):
                s.qvel[:2] += d * rng.uniform(0.3, 1.0)     # keep leaning into the wall (teacher forcing: any state is a valid start)
            pre = dict(qpos=s.qpos.copy(), qvel=s.qvel.copy(), qacc_ws=s.qacc_warmstart.copy(), goal=env.goal.copy(), action=a)
            s.min_activation_gap[0] = 1e30
            s.ctrl[:] = a.astype(np.float64)
            nwall = nfloor = 0
            for k in range(env.FRAME_SKIP):   # == sim.step(FRAME_SKIP) (orc_step is this loop); contacts of every substep are visible this way
                s.step(1)
                w = wall_contacts(s, wall_geoms)
                nwall += len(w) > 0
                nfloor += s.ncon > len(w)
            obs = env._obs()
            dist = np.linalg.norm(obs["achieved_goal"] - env.goal)
            for k2, v in pre.items():
                rec[k2].append(v)
            rec["obs"].append(obs["observation"]); rec["achieved"].append(obs["achieved_goal"]); rec["reward"].append(float(dist <= 0.45)); rec["success"].append(bool(dist <= 0.45))
            rec["qpos_next"].append(s.qpos.copy()); rec["qvel_next"].append(s.qvel.copy()); rec["ncon"].append(s.ncon); rec["nefc"].append(s.nefc)
            rec["wall_contact_substeps"].append(nwall); rec["floor_contact_substeps"].append(nfloor); rec["activation_gap"].append(float(s.min_activation_gap[0])); rec["segment"].append(seg)
            assert s.bad_state == 0
    out = {k: np.asarray(v) for k, v in rec.items()}
    path = os.path.join(ROOT, "tests", "golden", "ant_Large_teacher.npz")
    np.savez_compressed(path, **out)
    w, f = out["wall_contact_substeps"], out["floor_contact_substeps"]
    print(f"{len(w)} snapshots, {int((w > 0).sum())} with wall contacts in some substep ({int(((w > 0) & (f > 0)).sum())} with floor contacts as well), "
          f"max nefc {out['nefc'].max()}, max ncon {out['ncon'].max()}, max |qvel| {np.abs(out['qvel']).max():.1f}, {os.path.getsize(path) / 1024:.0f} KiB")
