"""Maze family specifications (host side): map tables, cell <-> xy mapping, wall injection, reset sampling.

Restates, in vectorisable form, the host logic of
  /root/reference/gymnasium_robotics/envs/maze/maps.py            (map tables, registry ids __init__.py:960-1078)
  /root/reference/gymnasium_robotics/envs/maze/maze_v4.py:148-242 (Maze.make_maze: walls + goal/reset cell lists)
  /root/reference/gymnasium_robotics/envs/maze/maze_v4.py:278-379 (generate_target_goal / generate_reset_pos / reset / add_xy_position_noise)
"""
import math
import xml.etree.ElementTree as ET

import numpy as np

R, G, C = "r", "g", "c"


def _parse(rows):
    """'1 0 r ...' strings -> list of lists with ints for 0/1 and 'r'/'g'/'c' markers"""
    return [[int(t) if t in "01" else t for t in row.split()] for row in rows]


MAPS = {
    "UMaze": _parse(["1 1 1 1 1", "1 0 0 0 1", "1 1 1 0 1", "1 0 0 0 1", "1 1 1 1 1"]),
    "Open": _parse(["1 1 1 1 1 1 1", "1 0 0 0 0 0 1", "1 0 0 0 0 0 1", "1 0 0 0 0 0 1", "1 1 1 1 1 1 1"]),
    "Open_Diverse_G": _parse(["1 1 1 1 1 1 1", "1 r g g g g 1", "1 g g g g g 1", "1 g g g g g 1", "1 1 1 1 1 1 1"]),
    "Open_Diverse_GR": _parse(["1 1 1 1 1 1 1", "1 c c c c c 1", "1 c c c c c 1", "1 c c c c c 1", "1 1 1 1 1 1 1"]),
    "Medium": _parse(["1 1 1 1 1 1 1 1", "1 0 0 1 1 0 0 1", "1 0 0 1 0 0 0 1", "1 1 0 0 0 1 1 1", "1 0 0 1 0 0 0 1", "1 0 1 0 0 1 0 1",
                      "1 0 0 0 1 0 0 1", "1 1 1 1 1 1 1 1"]),
    "Medium_Diverse_G": _parse(["1 1 1 1 1 1 1 1", "1 r 0 1 1 0 0 1", "1 0 0 1 0 0 g 1", "1 1 0 0 0 1 1 1", "1 0 0 1 0 0 0 1",
                                "1 g 1 0 0 1 0 1", "1 0 0 0 1 g 0 1", "1 1 1 1 1 1 1 1"]),
    "Medium_Diverse_GR": _parse(["1 1 1 1 1 1 1 1", "1 c 0 1 1 0 0 1", "1 0 0 1 0 0 c 1", "1 1 0 0 0 1 1 1", "1 0 0 1 0 0 0 1",
                                 "1 c 1 0 0 1 0 1", "1 0 0 0 1 c 0 1", "1 1 1 1 1 1 1 1"]),
    "Large": _parse(["1 1 1 1 1 1 1 1 1 1 1 1", "1 0 0 0 0 1 0 0 0 0 0 1", "1 0 1 1 0 1 0 1 0 1 0 1", "1 0 0 0 0 0 0 1 0 0 0 1",
                     "1 0 1 1 1 1 0 1 1 1 0 1", "1 0 0 1 0 1 0 0 0 0 0 1", "1 1 0 1 0 1 0 1 0 1 1 1", "1 0 0 1 0 0 0 1 0 0 0 1",
                     "1 1 1 1 1 1 1 1 1 1 1 1"]),
    "Large_Diverse_G": _parse(["1 1 1 1 1 1 1 1 1 1 1 1", "1 r 0 0 0 1 g 0 0 0 0 1", "1 0 1 1 0 1 0 1 0 1 0 1", "1 0 0 0 0 g 0 1 0 0 g 1",
                               "1 0 1 1 1 1 0 1 1 1 0 1", "1 0 g 1 0 1 0 0 0 0 0 1", "1 1 0 1 0 1 0 1 0 1 1 1", "1 0 0 1 g 0 g 1 0 g 0 1",
                               "1 1 1 1 1 1 1 1 1 1 1 1"]),
    "Large_Diverse_GR": _parse(["1 1 1 1 1 1 1 1 1 1 1 1", "1 c 0 0 0 1 c 0 0 0 0 1", "1 0 1 1 0 1 0 1 0 1 0 1", "1 0 0 0 0 c 0 1 0 0 c 1",
                                "1 0 1 1 1 1 0 1 1 1 0 1", "1 0 c 1 0 1 0 0 0 0 0 1", "1 1 0 1 0 1 0 1 0 1 1 1", "1 0 0 1 c 0 c 1 0 c 0 1",
                                "1 1 1 1 1 1 1 1 1 1 1 1"]),
}
POINT_MAX_EPISODE_STEPS = {"UMaze": 300, "Open": 300, "Medium": 600, "Large": 800}  # __init__.py:962-1078
POINT_MAZE_SIZE_SCALING, POINT_MAZE_HEIGHT = 1.0, 0.4                                   # point_maze.py:331-332
ANT_MAX_EPISODE_STEPS = {"UMaze": 700, "Open": 700, "Medium": 1000, "Large": 1000}    # __init__.py:839-958
ANT_MAZE_SIZE_SCALING, ANT_MAZE_HEIGHT, ANT_FRAME_SKIP = 4.0, 0.5, 5                    # ant_maze_v5.py:241-242; AntEnv frame_skip [3P]
GOAL_RADIUS = 0.45


def parse_point_maze_id(env_id: str):
    """'PointMaze_Medium_Diverse_GRDense-v3' -> ('Medium_Diverse_GR', 'dense', max_episode_steps)"""
    base = env_id.split("-v")[0]
    if not base.startswith("PointMaze_"):
        raise KeyError(f"unknown PointMaze env id {env_id}")
    name = base[len("PointMaze_"):]
    reward_type = "sparse"
    if name.endswith("Dense"):
        name, reward_type = name[: -len("Dense")], "dense"
    if name not in MAPS:
        raise KeyError(f"unknown PointMaze env id {env_id}")
    return name, reward_type, POINT_MAX_EPISODE_STEPS[name.split("_")[0]]


def parse_ant_maze_id(env_id: str):
    """'AntMaze_Large_Diverse_GR-v5' -> ('Large_Diverse_GR', 'sparse', 1000)"""
    base, _, ver = env_id.partition("-v")
    if not base.startswith("AntMaze_") or ver not in ("5",):
        raise KeyError(f"unknown / unsupported AntMaze env id {env_id} (only the v5 ids are in scope)")
    name = base[len("AntMaze_"):]
    reward_type = "sparse"
    if name.endswith("Dense"):
        name, reward_type = name[: -len("Dense")], "dense"
    if name not in MAPS:
        raise KeyError(f"unknown AntMaze env id {env_id}")
    return name, reward_type, ANT_MAX_EPISODE_STEPS[name.split("_")[0]]


class Maze:
    """Cell grid <-> simulation coordinates and the goal / reset cell lists (maze_v4.py:62-146,199-242)."""

    def __init__(self, maze_map, maze_size_scaling: float, maze_height: float):
        self.maze_map, self.maze_size_scaling, self.maze_height = maze_map, float(maze_size_scaling), float(maze_height)
        self.map_length, self.map_width = len(maze_map), len(maze_map[0])
        self.x_map_center = self.map_width / 2 * self.maze_size_scaling
        self.y_map_center = self.map_length / 2 * self.maze_size_scaling
        goals, resets, combined, empty, walls = [], [], [], [], []
        for i in range(self.map_length):
            for j in range(self.map_width):
                xy = self.cell_rowcol_to_xy((i, j))
                cell = maze_map[i][j]
                if cell == 1:
                    walls.append((i, j, xy))
                elif cell == R:
                    resets.append(xy)
                elif cell == G:
                    goals.append(xy)
                elif cell == C:
                    combined.append(xy)
                elif cell == 0:
                    empty.append(xy)
        if not goals and not resets and not combined:
            combined = empty
        elif not resets and not combined:
            resets = empty
        elif not goals and not combined:
            goals = empty
        self.unique_goal_locations = goals + combined
        self.unique_reset_locations = resets + combined
        self.walls = walls

    def cell_rowcol_to_xy(self, rowcol):
        x = (rowcol[1] + 0.5) * self.maze_size_scaling - self.x_map_center
        y = self.y_map_center - (rowcol[0] + 0.5) * self.maze_size_scaling
        return np.array([x, y])

    def add_walls(self, root: ET.Element) -> None:
        """MJCF rewrite of Maze.make_maze (maze_v4.py:168-212): one world-fixed box per wall cell + the target site."""
        wb = root.find("worldbody")
        s, h = self.maze_size_scaling, self.maze_height
        for i, j, xy in self.walls:
            ET.SubElement(wb, "geom", name=f"block_{i}_{j}", pos=f"{xy[0]} {xy[1]} {h / 2 * s}", size=f"{0.5 * s} {0.5 * s} {h / 2 * s}",
                          type="box", contype="1", conaffinity="1")
        ET.SubElement(wb, "site", name="target", pos=f"0 0 {h / 2 * s}", size=f"{0.2 * s}", type="sphere")


def sample_maze_reset(maze: Maze, rng, position_noise_range: float = 0.25, options=None):
    """PCG64 draw order of MazeEnv.reset (maze_v4.py:299-358).  Returns (goal_xy, reset_xy).
    Written on Python floats (the same float64 arithmetic as the reference's two-element arrays, without their per-call numpy
    overhead: this runs once per world and reset, thousands of times per vector reset)."""
    pr, sc = position_noise_range, maze.maze_size_scaling
    lo, span = -pr, pr - (-pr)

    def unif():      # Generator.uniform(low, high) is low + (high - low) * next_double: the same draw through the cheaper call
        return lo + span * rng.random()

    def check_cell(cell, what):
        assert maze.map_length > cell[0] and maze.map_width > cell[1]
        assert maze.maze_map[cell[0]][cell[1]] != 1, f"{what} can't be placed in a wall cell, {cell}"

    options = options or {}
    if options.get("goal_cell") is not None:
        check_cell(options["goal_cell"], "Goal")
        g = maze.cell_rowcol_to_xy(options["goal_cell"])
    else:
        g = maze.unique_goal_locations[rng.integers(0, len(maze.unique_goal_locations))]
    gx = float(g[0]) + unif() * sc
    gy = float(g[1]) + unif() * sc
    if options.get("reset_cell") is not None:
        check_cell(options["reset_cell"], "Reset")
        r = maze.cell_rowcol_to_xy(options["reset_cell"])
        rx, ry = float(r[0]), float(r[1])
    else:
        rx, ry, far, locs = gx, gy, 0.5 * sc, maze.unique_reset_locations
        while math.sqrt((rx - gx) * (rx - gx) + (ry - gy) * (ry - gy)) <= far:
            r = locs[rng.integers(0, len(locs))]
            rx, ry = float(r[0]), float(r[1])
    rx += unif() * sc
    ry += unif() * sc
    return np.array([gx, gy]), np.array([rx, ry])


def redraw_goal(maze: Maze, rng, achieved_xy, goal_xy, position_noise_range: float = 0.25):
    """MazeEnv.update_goal (maze_v4.py:400-418) for one world whose agent is within GOAL_RADIUS of its goal: draw goal cell +
    xy noise until the new goal is farther than the radius (same PCG64 draw order as the reference loop)."""
    goal = np.asarray(goal_xy, dtype=np.float64).copy()
    if len(maze.unique_goal_locations) <= 1:
        return goal
    achieved_xy = np.asarray(achieved_xy, dtype=np.float64)
    while np.linalg.norm(achieved_xy - goal) <= GOAL_RADIUS:
        goal = maze.unique_goal_locations[rng.integers(low=0, high=len(maze.unique_goal_locations))].copy()
        goal[0] += rng.uniform(low=-position_noise_range, high=position_noise_range) * maze.maze_size_scaling
        goal[1] += rng.uniform(low=-position_noise_range, high=position_noise_range) * maze.maze_size_scaling
    return goal
