"""MJCF -> flat model tables (host, cold path).

This is the replacement for ``mujoco.MjModel.from_xml_path`` at
/root/reference/gymnasium_robotics/envs/robot_env.py:293 for the MJCF subset
the five env families use.  It produces the tables
listed in ``include/grx_model_fields.def``:

* ``<include>``, nested ``<default class>`` / ``childclass``, ``<compiler>``
  angle / eulerseq / meshdir, ``<option>``;
* bodies (static bodies are fused into their moving ancestor -- dynamically
  equivalent because a joint-less body is welded to its parent), joints
  (free / slide / hinge), explicit ``<inertial>`` or geom-derived inertia;
* collidable geoms, sites, mocap bodies, weld equalities, contact excludes,
  joint-transmission actuators;
* compile-time constants the soft-constraint model needs: ``dof_invweight0``
  and per-body ``invweight0`` (inverse inertia seen at the body COM at
  ``qpos0``), ``meaninertia``;
* the statically filtered collision candidate list with mixed contact
  parameters.

Everything here is fp64 numpy; nothing on the per-step path calls it.
"""
from __future__ import annotations

import os
import struct
import xml.etree.ElementTree as ET
from dataclasses import dataclass, field
from typing import Dict, List, Optional

import numpy as np

from . import mathutil as mu

# MuJoCo enums (public API values)
JNT_FREE, JNT_BALL, JNT_SLIDE, JNT_HINGE = 0, 1, 2, 3
GEOM_PLANE, GEOM_HFIELD, GEOM_SPHERE, GEOM_CAPSULE, GEOM_ELLIPSOID, GEOM_CYLINDER, GEOM_BOX, GEOM_MESH = range(8)
GEOM_TYPES = {
    "plane": GEOM_PLANE,
    "hfield": GEOM_HFIELD,
    "sphere": GEOM_SPHERE,
    "capsule": GEOM_CAPSULE,
    "ellipsoid": GEOM_ELLIPSOID,
    "cylinder": GEOM_CYLINDER,
    "box": GEOM_BOX,
    "mesh": GEOM_MESH,
}
JNT_TYPES = {"free": JNT_FREE, "ball": JNT_BALL, "slide": JNT_SLIDE, "hinge": JNT_HINGE}
EQ_CONNECT, EQ_WELD, EQ_JOINT = 0, 1, 2
MJ_MINVAL = 1e-15

# dims / opt slot indices: keep in sync with include/grx_model.h
DIMS = [
    "nq", "nv", "nu", "nbody", "njnt", "ngeom", "nsite", "nmocap", "neq", "npair",
    "nmeshvert", "nmeshadj", "integrator", "iterations", "cone", "noslip_iterations",
    "eulerdamp", "ntree", "maxdepth", "maxefc_req", "jpool_req", "maxcon_req",
]
NDIMS = 32
OPTS = ["timestep", "gravity_x", "gravity_y", "gravity_z", "tolerance", "impratio", "meaninertia", "mpr_tolerance", "mpr_iterations", "noslip_tolerance",
        "origin_x", "origin_y", "origin_z"]
NOPTS = 16


def _floats(s, n=None, default=None):
    if s is None:
        return None if default is None else np.array(default, dtype=np.float64)
    v = np.array([float(x) for x in s.split()], dtype=np.float64)
    if n is not None and len(v) < n and default is not None:
        d = np.array(default, dtype=np.float64)
        d[: len(v)] = v
        v = d
    return v


def _bool(s, default=None):
    if s is None:
        return default
    return s.strip().lower() == "true"


# ----------------------------------------------------------------------------
# XML loading with <include> expansion
# ----------------------------------------------------------------------------
def _load_xml(path: str, base_dir: Optional[str] = None) -> ET.Element:
    root = ET.parse(path).getroot()
    base_dir = base_dir or os.path.dirname(os.path.abspath(path))
    _expand_includes(root, base_dir)
    return root


def _expand_includes(elem: ET.Element, base_dir: str) -> None:
    i = 0
    while i < len(elem):
        child = elem[i]
        if child.tag == "include":
            inc_path = os.path.join(base_dir, child.attrib["file"])
            inc_root = ET.parse(inc_path).getroot()
            # MuJoCo resolves nested includes relative to the top-level model dir
            _expand_includes(inc_root, base_dir)
            elem.remove(child)
            for k, sub in enumerate(list(inc_root)):
                elem.insert(i + k, sub)
            # do not advance: re-scan inserted nodes (already expanded)
            i += len(list(inc_root))
        else:
            _expand_includes(child, base_dir)
            i += 1


# ----------------------------------------------------------------------------
# defaults
# ----------------------------------------------------------------------------
class _Defaults:
    """Resolved attribute defaults per class name and element tag."""

    ACT_TAGS = ("general", "motor", "position", "velocity")

    def __init__(self):
        self.classes: Dict[str, Dict[str, Dict[str, str]]] = {"main": {}}

    def add_tree(self, elem: ET.Element, parent: str = None):
        if parent is None:
            name = elem.attrib.get("class", "main")
            cur = self.classes.setdefault(name, {})
        else:
            name = elem.attrib["class"]
            cur = {t: dict(a) for t, a in self.classes[parent].items()}
            self.classes[name] = cur
        for ch in elem:
            if ch.tag == "default":
                continue
            if ch.tag in self.ACT_TAGS:
                tag = "general"
                attrs = _actuator_shortcut(ch.tag, dict(ch.attrib)) if ch.tag != "general" else dict(ch.attrib)
            else:
                tag, attrs = ch.tag, dict(ch.attrib)
            cur.setdefault(tag, {}).update(attrs)
        for ch in elem:
            if ch.tag == "default":
                self.add_tree(ch, name)

    def get(self, cls: Optional[str], tag: str) -> Dict[str, str]:
        c = self.classes.get(cls or "main")
        if c is None:
            raise ValueError(f"unknown default class {cls}")
        return c.get(tag, {})


def _actuator_shortcut(tag: str, a: Dict[str, str]) -> Dict[str, str]:
    """Rewrite <position>/<motor>/<velocity> as <general> attributes."""
    a = dict(a)
    if tag == "motor":
        a.setdefault("gaintype", "fixed")
        a.setdefault("biastype", "none")
        a.setdefault("gainprm", "1 0 0")
    elif tag == "position":
        kp = float(a.pop("kp", "1")) if "kp" in a else None
        kv = float(a.pop("kv", "0")) if "kv" in a else None
        if kp is not None:
            a["gainprm"] = f"{kp} 0 0"
            a["_kp"] = str(kp)
        if kv is not None:
            a["_kv"] = str(kv)
        a["_position"] = "1"
    elif tag == "velocity":
        kv = float(a.pop("kv", "1"))
        a["gainprm"] = f"{kv} 0 0"
        a["biasprm"] = f"0 0 {-kv}"
        a["biastype"] = "affine"
    return a


# ----------------------------------------------------------------------------
# intermediate representation
# ----------------------------------------------------------------------------
@dataclass
class _Joint:
    name: str
    type: int
    pos: np.ndarray
    axis: np.ndarray
    ref: float
    springref: float
    range: np.ndarray
    limited: bool
    margin: float
    solref: np.ndarray
    solimp: np.ndarray
    stiffness: float
    damping: float
    armature: float
    frictionloss: float
    solreffriction: np.ndarray
    solimpfriction: np.ndarray
    body: int = -1
    qposadr: int = -1
    dofadr: int = -1


@dataclass
class _Geom:
    name: str
    type: int
    pos: np.ndarray
    quat: np.ndarray
    size: np.ndarray
    contype: int
    conaffinity: int
    condim: int
    priority: int
    friction: np.ndarray
    solmix: float
    solref: np.ndarray
    solimp: np.ndarray
    margin: float
    gap: float
    mass: Optional[float]
    density: float
    group: int
    mesh: Optional[str]
    body: int = -1


@dataclass
class _Site:
    name: str
    type: int
    pos: np.ndarray
    quat: np.ndarray
    size: np.ndarray
    body: int = -1


@dataclass
class _Body:
    name: str
    parent: int
    pos: np.ndarray
    quat: np.ndarray
    mocap: bool
    joints: List[_Joint] = field(default_factory=list)
    geoms: List[_Geom] = field(default_factory=list)
    sites: List[_Site] = field(default_factory=list)
    # inertial (explicit or derived)
    mass: float = 0.0
    ipos: np.ndarray = field(default_factory=lambda: np.zeros(3))
    inertia: np.ndarray = field(default_factory=lambda: np.zeros((3, 3)))  # about COM, body frame
    explicit_inertial: bool = False


@dataclass
class CompiledModel:
    """Flat tables (dict of numpy arrays) + name maps + blob packing."""

    tables: Dict[str, np.ndarray]
    names: Dict[str, Dict[str, int]]
    info: Dict[str, object]

    def dim(self, key: str) -> int:
        return int(self.tables["dims"][DIMS.index(key)])

    def opt(self, key: str) -> float:
        return float(self.tables["opt"][OPTS.index(key)])

    def pack(self):
        return pack_blob(self.tables)

    @property
    def origin(self) -> np.ndarray:
        """MJCF coordinates of the compiled model's world origin (compile_mjcf(origin=...)); zeros for a model compiled in the MJCF's own frame
        (blobs written before the field existed carry zeros in these opt slots)."""
        o = self.tables["opt"]
        return np.array([o[OPTS.index("origin_x")], o[OPTS.index("origin_y")], o[OPTS.index("origin_z")]], dtype=np.float64)

    # ---- frames.  A model compiled with origin != 0 lives in a translated world frame; these helpers are the ONLY place that knows which words of a state row are world positions.
    def world_columns(self, key: str):
        """(columns, axes) of the state row `key` ("qpos" | "mocap" | "aux") that hold a world position: the translation of every free joint whose body hangs on the world,
        the position triple of every mocap body.  Other rows (qvel, warm start, ctrl, goals, ...) carry no absolute position."""
        t = self.tables
        cols, axes = [], []
        if key == "qpos":
            jt, ja, jb = np.asarray(t["jnt_type"]).reshape(-1), np.asarray(t["jnt_qposadr"]).reshape(-1), np.asarray(t["jnt_bodyid"]).reshape(-1)
            par = np.asarray(t["body_parent"]).reshape(-1)
            for j in range(len(jt)):
                if int(jt[j]) == 0 and int(par[int(jb[j])]) == 0:
                    cols += [int(ja[j]) + e for e in range(3)]
                    axes += [0, 1, 2]
        elif key == "mocap":
            for k in range(self.dim("nmocap")):
                cols += [7 * k + e for e in range(3)]
                axes += [0, 1, 2]
        elif key == "aux" and self.dim("nmocap"):      # Fetch: pose of gripper_link at the last forward pass (csrc/grx_fetch_task.h GrxFetchBuffers::aux), position first
            cols, axes = [0, 1, 2], [0, 1, 2]
        return np.asarray(cols, dtype=np.int64), np.asarray(axes, dtype=np.int64)

    def rows_from_world(self, key: str, values) -> np.ndarray:
        """state rows given in the MJCF's world frame (what the reference / the oracle / a fixture holds) -> the same rows in THIS model's frame, fp64 (the caller rounds to fp32)"""
        out = np.array(values, dtype=np.float64)
        cols, axes = self.world_columns(key)
        if len(cols) and self.origin.any():
            out[..., cols] -= self.origin[axes]
        return out

    def rows_to_world(self, key: str, rows) -> np.ndarray:
        """inverse of rows_from_world, fp64"""
        out = np.array(rows, dtype=np.float64)
        cols, axes = self.world_columns(key)
        if len(cols) and self.origin.any():
            out[..., cols] += self.origin[axes]
        return out

    def in_mjcf_frame(self) -> "CompiledModel":
        """The same model in the MJCF's own world frame (origin 0): what hangs directly on the world is translated back.  The oracle runs THIS (oracle/oracle_sim.py), so that the
        checker keeps restating the reference on the reference's own coordinates; tests/test_cpu_origin.py pins it against a compilation with origin 0."""
        o = self.origin
        if not o.any():
            return self
        m = self.copy()
        t = m.tables
        par = np.asarray(t["body_parent"]).reshape(-1)
        bp = np.array(t["body_pos"], dtype=np.float64).reshape(-1, 3)
        bp[1:][par[1:] == 0] += o
        t["body_pos"] = bp.reshape(np.asarray(t["body_pos"]).shape)
        for pos, bid in (("geom_pos", "geom_bodyid"), ("site_pos", "site_bodyid")):
            a = np.array(t[pos], dtype=np.float64).reshape(-1, 3)
            if len(a):
                a[np.asarray(t[bid]).reshape(-1) == 0] += o
            t[pos] = a.reshape(np.asarray(t[pos]).shape)
        if "mocap_pos0" in t and np.asarray(t["mocap_pos0"]).size:
            t["mocap_pos0"] = np.array(t["mocap_pos0"], dtype=np.float64) + o
        q0 = np.array(t["qpos0"], dtype=np.float64)
        cols, axes = self.world_columns("qpos")
        q0[cols] += o[axes]
        t["qpos0"] = q0
        for k, name in enumerate(("origin_x", "origin_y", "origin_z")):
            t["opt"][OPTS.index(name)] = 0.0
        m.info["origin"] = [0.0, 0.0, 0.0]
        return m

    def with_capacity(self, **capacity) -> "CompiledModel":
        """Copy of the model with other engine table capacities (maxcon / maxefc / jpool; 0 = engine default).  The capacities are
        requests stored in the dims block, so no recompilation is needed; a model whose capacities match no specialised kernel runs on
        the generic one."""
        m = self.copy()
        for key, val in capacity.items():
            m.tables["dims"][DIMS.index(key + "_req")] = int(val)
        return m

    def copy(self) -> "CompiledModel":
        return CompiledModel({k: v.copy() for k, v in self.tables.items()},
                             {k: dict(v) for k, v in self.names.items()}, dict(self.info))


# ----------------------------------------------------------------------------
# field list (parsed from the shared .def so Python and C cannot drift)
# ----------------------------------------------------------------------------
_DEF_PATH = os.path.join(os.path.dirname(__file__), "..", "..", "include", "grx_model_fields.def")
_PKG_DEF_PATH = os.path.join(os.path.dirname(__file__), "grx_model_fields.def")


def field_list():
    path = _DEF_PATH if os.path.exists(_DEF_PATH) else _PKG_DEF_PATH
    out = []
    with open(path) as f:
        for line in f:
            line = line.strip()
            if line.startswith("GRX_FI(") or line.startswith("GRX_FF("):
                kind = "i" if line.startswith("GRX_FI(") else "f"
                name = line[line.index("(") + 1: line.index(")")].strip()
                out.append((name, kind))
    return out


def pack_blob(tables: Dict[str, np.ndarray]):
    """-> (H int32[2*nfields], I int32[], F float64[]) as documented in the .def"""
    fl = field_list()
    H = np.zeros(2 * len(fl), dtype=np.int32)
    ints, flts = [], []
    io = fo = 0
    for k, (name, kind) in enumerate(fl):
        if name not in tables and name in ("devpair_gate", "gate_qadr", "gate_box"):      # blobs written before the pair gates existed: no gates
            tables = dict(tables, devpair_gate=-np.ones(np.asarray(tables["devpair"]).size, np.int32), gate_qadr=np.zeros(0, np.int32), gate_box=np.zeros(0))
        a = np.ascontiguousarray(tables[name]).ravel()
        if kind == "i":
            a = a.astype(np.int32)
            H[2 * k], H[2 * k + 1] = io, a.size
            ints.append(a)
            io += a.size
        else:
            a = a.astype(np.float64)
            H[2 * k], H[2 * k + 1] = fo, a.size
            flts.append(a)
            fo += a.size
    I = np.concatenate(ints) if ints else np.zeros(0, np.int32)
    F = np.concatenate(flts) if flts else np.zeros(0, np.float64)
    return H, np.ascontiguousarray(I, dtype=np.int32), np.ascontiguousarray(F, dtype=np.float64)


def save_model(model: CompiledModel, path: str) -> None:
    import json

    np.savez_compressed(
        path,
        __names__=np.frombuffer(json.dumps(model.names).encode(), dtype=np.uint8),
        __info__=np.frombuffer(json.dumps(model.info, default=str).encode(), dtype=np.uint8),
        **model.tables,
    )


def load_model(path: str) -> CompiledModel:
    import json

    z = np.load(path)
    names = json.loads(bytes(z["__names__"]).decode())
    info = json.loads(bytes(z["__info__"]).decode())
    tables = {k: z[k] for k in z.files if not k.startswith("__")}
    return CompiledModel(tables, names, info)


# ----------------------------------------------------------------------------
# the compiler
# ----------------------------------------------------------------------------
class MjcfCompiler:
    def __init__(self, xml_path: str, mutate=None, capacity=None, touch_filter=None, keep_sites=None, shift_body=None, shift_rotates=False, origin=None):
        """mutate: optional callable(root_element) applied after <include> expansion, before compilation -- the
        in-memory equivalent of the reference's Maze.make_maze XML rewrite (envs/maze/maze_v4.py:168-242).
        origin: world point (MJCF coordinates) that becomes the origin of the COMPILED model's world frame: every direct child of the
        worldbody (bodies incl. free and mocap bodies, static geoms, sites) is translated by -origin.  Rigid-body physics is translation
        invariant, fp32 is not: the Shadow hand sits at (1, 1.25, 0.15) in its MJCF (assets/hand/robot.xml:3), where an fp32 position carries
        an ulp of 1.2e-7 m; in a palm-centred frame the same positions carry 1.5e-8 .. 3e-8.  The value is stored in the model
        (opt origin_x/y/z, info["origin"]); the task layers add it back in fp64 wherever a world position leaves the engine."""
        self.origin = np.zeros(3) if origin is None else np.asarray(origin, dtype=np.float64).reshape(3).copy()
        self.xml_path = os.path.abspath(xml_path)
        self.dir = os.path.dirname(self.xml_path)
        self.root = _load_xml(self.xml_path)
        if mutate is not None:
            mutate(self.root)
        self.keep_sites = None if keep_sites is None else set(keep_sites)   # names of the sites the engine tracks (None = all)
        self.touch_filter = touch_filter      # callable(sensor name) -> bool: which <touch> sensors the engine evaluates
        # name of a static child of the world whose position is per-world STATE (the reference rewrites model.body_pos at reset:
        # adroit_hammer.py:374-376): everything welded to it gets a shift flag, the engine adds the world's shift vector to it
        # A shift body WITH joints (adroit_relocate.py:358-363 moves the ball's body) only offsets its own origin.  shift_rotates: the group's
        # members are also rotated about the world origin by the world's quaternion (adroit_pen.py:381 rewrites model.body_quat of the target).
        self.shift_body = shift_body
        self.shift_rotates = bool(shift_rotates)
        self.capacity = dict(capacity or {})   # engine row-table / Jacobian-pool capacities requested for this model (0 = default)
        self.defaults = _Defaults()
        self.angle_scale = np.pi / 180.0  # MJCF default angle unit is degree
        self.eulerseq = "xyz"
        self.meshdir = ""
        self.autolimits = True
        self.inertiafromgeom = "auto"
        self.inertiagrouprange = (0, 5)
        self.bodies: List[_Body] = []
        self.meshes: Dict[str, Dict[str, object]] = {}
        self.opt = dict(timestep=0.002, gravity=np.array([0, 0, -9.81]), tolerance=1e-8, impratio=1.0,
                        integrator=0, iterations=100, cone=0, noslip_iterations=0, eulerdamp=1, mpr_tolerance=1e-6, mpr_iterations=50,
                        noslip_tolerance=1e-6)

    # -- attribute helpers -------------------------------------------------
    def _attrs(self, elem: ET.Element, tag: str, childclass: Optional[str]) -> Dict[str, str]:
        cls = elem.attrib.get("class", childclass)
        a = dict(self.defaults.get(cls, tag))
        a.update(elem.attrib)
        return a

    def _orientation(self, a: Dict[str, str]) -> np.ndarray:
        if "quat" in a:
            return mu.quat_normalize(_floats(a["quat"]))
        if "euler" in a:
            return mu.euler2quat(_floats(a["euler"]) * self.angle_scale, self.eulerseq)
        if "axisangle" in a:
            v = _floats(a["axisangle"])
            return mu.axisangle2quat(v[:3], v[3] * self.angle_scale)
        if "xyaxes" in a:
            return mu.xyaxes2quat(_floats(a["xyaxes"]))
        if "zaxis" in a:
            return mu.zaxis2quat(_floats(a["zaxis"]))
        return np.array([1.0, 0.0, 0.0, 0.0])

    # -- sections ------------------------------------------------------------
    def _parse_globals(self):
        for c in self.root.findall("compiler"):
            a = c.attrib
            if "angle" in a:
                self.angle_scale = 1.0 if a["angle"] == "radian" else np.pi / 180.0
            self.eulerseq = a.get("eulerseq", self.eulerseq)
            self.meshdir = a.get("meshdir", self.meshdir)
            if "autolimits" in a:
                self.autolimits = _bool(a["autolimits"])
            self.inertiafromgeom = a.get("inertiafromgeom", self.inertiafromgeom).lower()
            if "inertiagrouprange" in a:
                g = a["inertiagrouprange"].split()
                self.inertiagrouprange = (int(g[0]), int(g[1]))
            if a.get("coordinate", "local") != "local":
                raise NotImplementedError("coordinate=global")
        for o in self.root.findall("option"):
            a = o.attrib
            if "timestep" in a:
                self.opt["timestep"] = float(a["timestep"])
            if "gravity" in a:
                self.opt["gravity"] = _floats(a["gravity"])
            if "tolerance" in a:
                self.opt["tolerance"] = float(a["tolerance"])
            if "impratio" in a:
                self.opt["impratio"] = float(a["impratio"])
            if "iterations" in a:
                self.opt["iterations"] = int(a["iterations"])
            if "noslip_iterations" in a:
                self.opt["noslip_iterations"] = int(a["noslip_iterations"])
            if "noslip_tolerance" in a:
                self.opt["noslip_tolerance"] = float(a["noslip_tolerance"])
            if "mpr_tolerance" in a:
                self.opt["mpr_tolerance"] = float(a["mpr_tolerance"])
            if "mpr_iterations" in a:
                self.opt["mpr_iterations"] = int(a["mpr_iterations"])
            if "integrator" in a:
                self.opt["integrator"] = {"euler": 0, "rk4": 1, "implicit": 2, "implicitfast": 3}[a["integrator"].lower()]
            if "cone" in a:
                self.opt["cone"] = {"pyramidal": 0, "elliptic": 1}[a["cone"].lower()]
            for fl in o.findall("flag"):
                if fl.attrib.get("eulerdamp", "enable") == "disable":
                    self.opt["eulerdamp"] = 0
        for d in self.root.findall("default"):
            self.defaults.add_tree(d)
        for asset in self.root.findall("asset"):
            for m in asset.findall("mesh"):
                a = dict(self.defaults.get(m.attrib.get("class"), "mesh"))
                a.update(m.attrib)
                name = a.get("name") or os.path.splitext(os.path.basename(a["file"]))[0]
                self.meshes[name] = dict(file=a["file"], scale=_floats(a.get("scale"), 3, [1, 1, 1]))

    # -- bodies ------------------------------------------------------------------
    def _parse_body(self, elem: ET.Element, parent: int, childclass: Optional[str]):
        a = elem.attrib
        childclass = a.get("childclass", childclass)
        body = _Body(
            name=a.get("name", f"_body{len(self.bodies)}"),
            parent=parent,
            pos=_floats(a.get("pos"), 3, [0, 0, 0]),
            quat=self._orientation(a),
            mocap=_bool(a.get("mocap"), False),
        )
        bid = len(self.bodies)
        self.bodies.append(body)
        for ch in elem:
            if ch.tag == "inertial":
                ia = ch.attrib
                body.explicit_inertial = True
                body.mass = float(ia["mass"])
                body.ipos = _floats(ia.get("pos"), 3, [0, 0, 0])
                iq = self._orientation(ia)
                if "fullinertia" in ia:
                    f = _floats(ia["fullinertia"])
                    I = np.array([[f[0], f[3], f[4]], [f[3], f[1], f[5]], [f[4], f[5], f[2]]])
                else:
                    I = np.diag(_floats(ia.get("diaginertia"), 3, [0, 0, 0]))
                R = mu.quat2mat(iq)
                body.inertia = R @ I @ R.T
            elif ch.tag in ("joint", "freejoint"):
                body.joints.append(self._parse_joint(ch, childclass, bid))
            elif ch.tag == "geom":
                body.geoms.append(self._parse_geom(ch, childclass, bid))
            elif ch.tag == "site":
                body.sites.append(self._parse_site(ch, childclass, bid))
        for ch in elem:
            if ch.tag == "body":
                self._parse_body(ch, bid, childclass)
        return bid

    def _parse_joint(self, elem, childclass, bid) -> _Joint:
        if elem.tag == "freejoint":
            a = dict(elem.attrib)
            a["type"] = "free"
        else:
            a = self._attrs(elem, "joint", childclass)
        jt = JNT_TYPES[a.get("type", "hinge")]
        ang = self.angle_scale if jt in (JNT_HINGE, JNT_BALL) else 1.0
        rng = _floats(a.get("range"), 2, [0, 0])
        if "limited" in a and a["limited"] != "auto":
            limited = _bool(a["limited"])
        else:
            limited = bool(self.autolimits and "range" in a)
        axis = _floats(a.get("axis"), 3, [0, 0, 1])
        n = np.linalg.norm(axis)
        axis = axis / n if n > 0 else axis
        ref = float(a.get("ref", 0.0)) * ang
        return _Joint(
            name=a.get("name", f"_jnt_{bid}_{id(elem)}"),
            type=jt,
            pos=_floats(a.get("pos"), 3, [0, 0, 0]),
            axis=axis,
            ref=ref,
            springref=float(a.get("springref", 0.0)) * ang,
            range=rng * ang,
            limited=limited and jt in (JNT_SLIDE, JNT_HINGE, JNT_BALL),
            margin=float(a.get("margin", 0.0)),
            solref=_floats(a.get("solreflimit"), 2, [0.02, 1.0]),
            solimp=_floats(a.get("solimplimit"), 5, [0.9, 0.95, 0.001, 0.5, 2.0]),
            stiffness=float(a.get("stiffness", 0.0)),
            damping=float(a.get("damping", 0.0)),
            armature=float(a.get("armature", 0.0)),
            frictionloss=float(a.get("frictionloss", 0.0)),
            solreffriction=_floats(a.get("solreffriction"), 2, [0.02, 1.0]),
            solimpfriction=_floats(a.get("solimpfriction"), 5, [0.9, 0.95, 0.001, 0.5, 2.0]),
            body=bid,
        )

    def _parse_geom(self, elem, childclass, bid) -> _Geom:
        a = self._attrs(elem, "geom", childclass)
        gt = GEOM_TYPES[a.get("type", "sphere")]
        size = _floats(a.get("size"), 3, [0, 0, 0])
        pos = _floats(a.get("pos"), 3, [0, 0, 0])
        quat = self._orientation(a)
        if "fromto" in a:
            ft = _floats(a["fromto"])
            p0, p1 = ft[:3], ft[3:]
            pos = 0.5 * (p0 + p1)
            quat = mu.zaxis2quat(p1 - p0)
            half = 0.5 * np.linalg.norm(p1 - p0)
            if gt in (GEOM_CAPSULE, GEOM_CYLINDER):
                size = np.array([size[0], half, 0.0])
            elif gt in (GEOM_BOX, GEOM_ELLIPSOID):
                size = np.array([size[0], size[0], half])
        mass = float(a["mass"]) if "mass" in a else None
        return _Geom(
            name=a.get("name", ""),
            type=gt,
            pos=pos,
            quat=quat,
            size=size,
            contype=int(a.get("contype", 1)),
            conaffinity=int(a.get("conaffinity", 1)),
            condim=int(a.get("condim", 3)),
            priority=int(a.get("priority", 0)),
            friction=_floats(a.get("friction"), 3, [1.0, 0.005, 0.0001]),
            solmix=float(a.get("solmix", 1.0)),
            solref=_floats(a.get("solref"), 2, [0.02, 1.0]),
            solimp=_floats(a.get("solimp"), 5, [0.9, 0.95, 0.001, 0.5, 2.0]),
            margin=float(a.get("margin", 0.0)),
            gap=float(a.get("gap", 0.0)),
            mass=mass,
            density=float(a.get("density", 1000.0)),
            group=int(a.get("group", 0)),
            mesh=a.get("mesh"),
            body=bid,
        )

    def _parse_site(self, elem, childclass, bid) -> _Site:
        a = self._attrs(elem, "site", childclass)
        st = GEOM_TYPES[a.get("type", "sphere")]
        size = _floats(a.get("size"), 3, [0.005, 0.005, 0.005])
        pos = _floats(a.get("pos"), 3, [0, 0, 0])
        quat = self._orientation(a)
        if "fromto" in a:
            ft = _floats(a["fromto"])
            pos = 0.5 * (ft[:3] + ft[3:])
            quat = mu.zaxis2quat(ft[3:] - ft[:3])
            size = np.array([size[0], 0.5 * np.linalg.norm(ft[3:] - ft[:3]), 0.0])
        return _Site(name=a.get("name", ""), type=st, pos=pos, quat=quat, size=size, body=bid)

    # -- mesh hulls -----------------------------------------------------------------
    def _load_mesh_hull(self, name: str):
        """Returns (hull vertices (n,3), adjacency list) of the mesh's convex hull."""
        m = self.meshes[name]
        if "hull" in m:
            return m["hull"]
        path = os.path.join(self.dir, self.meshdir, m["file"])
        with open(path, "rb") as f:
            b = f.read()
        ntri = struct.unpack("<I", b[80:84])[0]
        if 84 + 50 * ntri != len(b):
            raise NotImplementedError(f"only binary STL supported: {path}")
        rec = np.frombuffer(b[84: 84 + 50 * ntri], dtype=np.uint8).reshape(ntri, 50)
        v = rec[:, 12:48].copy().view("<f4").reshape(-1, 3).astype(np.float64) * m["scale"]
        v = np.unique(v, axis=0)
        from scipy.spatial import ConvexHull

        hull = ConvexHull(v)
        idx = np.unique(hull.simplices.ravel())
        remap = -np.ones(len(v), dtype=np.int64)
        remap[idx] = np.arange(len(idx))
        hv = v[idx]
        adj = [set() for _ in range(len(idx))]
        for s in hull.simplices:
            for i in range(3):
                p, q = remap[s[i]], remap[s[(i + 1) % 3]]
                adj[p].add(int(q))
                adj[q].add(int(p))
        m["hull"] = (hv, [sorted(s) for s in adj])
        return m["hull"]

    def _mesh_center(self, name: str):
        """Centre of mass of the mesh's convex hull (mesh frame): the origin of the compiled geom frame."""
        m = self.meshes[name]
        if "ctr" not in m:
            m["ctr"] = _polyhedron_mass_props(self._load_mesh_hull(name)[0])[1]
        return m["ctr"]

    # -- inertia from geoms -----------------------------------------------------------
    @staticmethod
    def _geom_volume_inertia(g: _Geom):
        """(volume, unit-density inertia about geom centre in geom frame) for primitives."""
        s = g.size
        if g.type == GEOM_SPHERE:
            vol = 4.0 / 3.0 * np.pi * s[0] ** 3
            I = np.eye(3) * (0.4 * vol * s[0] ** 2)
        elif g.type == GEOM_BOX:
            vol = 8 * s[0] * s[1] * s[2]
            I = np.diag([vol / 3 * (s[1] ** 2 + s[2] ** 2), vol / 3 * (s[0] ** 2 + s[2] ** 2), vol / 3 * (s[0] ** 2 + s[1] ** 2)])
        elif g.type == GEOM_CYLINDER:
            r, h = s[0], s[1]
            vol = np.pi * r * r * 2 * h
            ix = vol * (3 * r * r + (2 * h) ** 2) / 12.0
            I = np.diag([ix, ix, vol * r * r / 2])
        elif g.type == GEOM_CAPSULE:
            r, h = s[0], s[1]
            vc = np.pi * r * r * 2 * h
            vs = 4.0 / 3.0 * np.pi * r ** 3
            vol = vc + vs
            ixc = vc * (3 * r * r + (2 * h) ** 2) / 12.0
            izc = vc * r * r / 2
            # two hemispheres: sphere inertia + parallel axis for hemisphere offsets
            izs = 0.4 * vs * r * r
            ixs = 0.4 * vs * r * r + vs * h * h + 0.75 * vs * r * h
            I = np.diag([ixc + ixs, ixc + ixs, izc + izs])
        elif g.type == GEOM_ELLIPSOID:
            vol = 4.0 / 3.0 * np.pi * s[0] * s[1] * s[2]
            I = np.diag([vol / 5 * (s[1] ** 2 + s[2] ** 2), vol / 5 * (s[0] ** 2 + s[2] ** 2), vol / 5 * (s[0] ** 2 + s[1] ** 2)])
        else:
            return 0.0, np.zeros((3, 3))
        return vol, I

    def _derive_inertia(self, body: _Body):
        if self.inertiafromgeom == "false" or (self.inertiafromgeom == "auto" and body.explicit_inertial):
            return
        tot_m = 0.0
        com = np.zeros(3)
        parts = []
        for g in body.geoms:
            if not (self.inertiagrouprange[0] <= g.group <= self.inertiagrouprange[1]):
                continue
            if g.type == GEOM_MESH:
                hv, _ = self._load_mesh_hull(g.mesh)
                vol, c, I = _polyhedron_mass_props(hv)
                gm = g.mass if g.mass is not None else g.density * vol
                scale = gm / vol if vol > 0 else 0.0
                R = mu.quat2mat(g.quat)
                parts.append((gm, g.pos + R @ c, R @ (I * scale) @ R.T))
            else:
                vol, I = self._geom_volume_inertia(g)
                if vol <= 0:
                    continue
                gm = g.mass if g.mass is not None else g.density * vol
                R = mu.quat2mat(g.quat)
                parts.append((gm, g.pos.copy(), R @ (I * gm / vol) @ R.T))
        for gm, c, _ in parts:
            tot_m += gm
            com += gm * c
        if tot_m <= 0:
            return
        com /= tot_m
        I = np.zeros((3, 3))
        for gm, c, Ig in parts:
            d = c - com
            I += Ig + gm * (np.dot(d, d) * np.eye(3) - np.outer(d, d))
        body.mass, body.ipos, body.inertia = tot_m, com, I

    # -- main ------------------------------------------------------------------------
    def compile(self) -> CompiledModel:
        self._parse_globals()
        world = _Body(name="world", parent=-1, pos=np.zeros(3), quat=np.array([1.0, 0, 0, 0]), mocap=False)
        self.bodies.append(world)
        for wb in self.root.findall("worldbody"):
            for ch in wb:
                if ch.tag == "geom":
                    world.geoms.append(self._parse_geom(ch, None, 0))
                elif ch.tag == "site":
                    world.sites.append(self._parse_site(ch, None, 0))
            for ch in wb:
                if ch.tag == "body":
                    self._parse_body(ch, 0, None)
        if self.origin.any():      # the compiled world frame = the MJCF's translated by -origin: only what hangs directly on the world carries an absolute position
            if self.shift_rotates:
                raise NotImplementedError("origin with a rotating shift group (the group turns about the MJCF's world origin)")
            for g in world.geoms:
                g.pos = g.pos - self.origin
            for st in world.sites:
                st.pos = st.pos - self.origin
            for b in self.bodies[1:]:
                if b.parent == 0:
                    b.pos = b.pos - self.origin
        for b in self.bodies[1:]:
            self._derive_inertia(b)
        return _Lowering(self).run()


def _polyhedron_mass_props(hv: np.ndarray):
    """Volume, COM and unit-density inertia (about COM) of the convex hull of hv."""
    from scipy.spatial import ConvexHull

    hull = ConvexHull(hv)
    c0 = hv.mean(axis=0)
    vol = 0.0
    com = np.zeros(3)
    C = np.zeros((3, 3))  # integral of x x^T about c0
    for s in hull.simplices:
        a, b, c = hv[s[0]] - c0, hv[s[1]] - c0, hv[s[2]] - c0
        v = abs(np.dot(a, np.cross(b, c))) / 6.0
        vol += v
        com += v * (a + b + c) / 4.0
        sm = a + b + c
        C += v / 20.0 * (np.outer(a, a) + np.outer(b, b) + np.outer(c, c) + np.outer(sm, sm))
    com /= vol
    C = C - vol * np.outer(com, com)
    I = np.trace(C) * np.eye(3) - C
    return vol, c0 + com, I


# ----------------------------------------------------------------------------
# lowering: full tree -> invweights -> fused tree -> tables
# ----------------------------------------------------------------------------
class _Lowering:
    def __init__(self, c: MjcfCompiler):
        self.c = c
        self.bodies = c.bodies

    # full-tree forward kinematics at qpos0 (all joints at their reference)
    def _kin0(self):
        nb = len(self.bodies)
        xpos = np.zeros((nb, 3))
        xquat = np.zeros((nb, 4))
        xquat[0] = [1, 0, 0, 0]
        for i in range(1, nb):
            b = self.bodies[i]
            p = b.parent
            xpos[i] = xpos[p] + mu.rot_vec(xquat[p], b.pos)
            xquat[i] = mu.quat_normalize(mu.quat_mul(xquat[p], b.quat))
        return xpos, xquat

    def run(self) -> CompiledModel:
        c = self.c
        B = self.bodies
        nb = len(B)
        xpos, xquat = self._kin0()
        xmat = np.array([mu.quat2mat(q) for q in xquat])

        # ---- dof enumeration on the full tree
        jnts: List[_Joint] = []
        nq = nv = 0
        for bi, b in enumerate(B):
            for j in b.joints:
                j.body = bi
                j.qposadr, j.dofadr = nq, nv
                nq += {JNT_FREE: 7, JNT_BALL: 4}.get(j.type, 1)
                nv += {JNT_FREE: 6, JNT_BALL: 3}.get(j.type, 1)
                jnts.append(j)
        weld = list(range(nb))  # weld id: nearest ancestor-or-self with joints (0 = static)
        for i in range(1, nb):
            if not B[i].joints:
                weld[i] = weld[B[i].parent] if not B[i].mocap else 0
        # dof -> (joint, local index)
        dof_jnt = []
        for j in jnts:
            dof_jnt += [(j, k) for k in range({JNT_FREE: 6, JNT_BALL: 3}.get(j.type, 1))]

        def chain_dofs(bi):
            """dofs affecting body bi, root first"""
            out = []
            while bi > 0:
                for j in reversed(B[bi].joints):
                    n = {JNT_FREE: 6, JNT_BALL: 3}.get(j.type, 1)
                    out = list(range(j.dofadr, j.dofadr + n)) + out
                bi = B[bi].parent
            return out

        def jac(bi, point):
            """6 x nv (translation rows first) Jacobian of a world point fixed to body bi at qpos0"""
            J = np.zeros((6, nv))
            for d in chain_dofs(bi):
                j, k = dof_jnt[d]
                jb = j.body
                R = xmat[jb]
                if j.type == JNT_SLIDE:
                    J[:3, d] = R @ j.axis
                elif j.type == JNT_HINGE:
                    ax = R @ j.axis
                    anchor = xpos[jb] + R @ j.pos
                    J[:3, d] = np.cross(ax, point - anchor)
                    J[3:, d] = ax
                elif j.type == JNT_FREE:
                    if k < 3:
                        J[k, d] = 1.0
                    else:
                        ax = R[:, k - 3]
                        J[:3, d] = np.cross(ax, point - xpos[jb])
                        J[3:, d] = ax
                else:
                    ax = R[:, k]
                    anchor = xpos[jb] + R @ j.pos
                    J[:3, d] = np.cross(ax, point - anchor)
                    J[3:, d] = ax
            return J

        # ---- mass matrix at qpos0 (sum_b J^T I J + armature)
        M = np.zeros((nv, nv))
        for bi in range(1, nb):
            b = B[bi]
            if b.mass <= 0 and not np.any(b.inertia):
                continue
            xi = xpos[bi] + xmat[bi] @ b.ipos
            J = jac(bi, xi)
            Iw = xmat[bi] @ b.inertia @ xmat[bi].T
            M += b.mass * J[:3].T @ J[:3] + J[3:].T @ Iw @ J[3:]
        for d, (j, k) in enumerate(dof_jnt):
            M[d, d] += j.armature
        Minv = np.linalg.inv(M) if nv else np.zeros((0, 0))
        dof_invweight0 = np.diag(Minv).copy() if nv else np.zeros(0)
        for j in jnts:
            if j.type == JNT_FREE:
                a = j.dofadr
                dof_invweight0[a: a + 3] = dof_invweight0[a: a + 3].mean()
                dof_invweight0[a + 3: a + 6] = dof_invweight0[a + 3: a + 6].mean()
            elif j.type == JNT_BALL:
                a = j.dofadr
                dof_invweight0[a: a + 3] = dof_invweight0[a: a + 3].mean()
        body_invweight0 = np.zeros((nb, 2))
        for bi in range(1, nb):
            if weld[bi] == 0:
                continue
            xi = xpos[bi] + xmat[bi] @ B[bi].ipos
            J = jac(bi, xi)
            A = J @ Minv @ J.T
            body_invweight0[bi, 0] = max(MJ_MINVAL, np.trace(A[:3, :3]) / 3)
            body_invweight0[bi, 1] = max(MJ_MINVAL, np.trace(A[3:, 3:]) / 3)
        meaninertia = float(np.mean(np.diag(M))) if nv else 1.0

        # ---- fuse static bodies
        keep = [i for i in range(nb) if i == 0 or B[i].joints or B[i].mocap]
        new_id = {old: k for k, old in enumerate(keep)}

        def anchor_of(i):
            """(kept ancestor-or-self, relative pos, relative quat) of body i"""
            p = np.zeros(3)
            q = np.array([1.0, 0, 0, 0])
            while i not in new_id:
                b = B[i]
                p = b.pos + mu.rot_vec(b.quat, p)
                q = mu.quat_mul(b.quat, q)
                i = b.parent
            return i, p, mu.quat_normalize(q)

        fb_mass = {k: 0.0 for k in keep}
        fb_com = {k: np.zeros(3) for k in keep}
        parts = {k: [] for k in keep}
        for i in range(1, nb):
            b = B[i]
            if b.mass <= 0 and not np.any(b.inertia):
                continue
            k, p, q = anchor_of(i)
            if k == 0:
                continue
            R = mu.quat2mat(q)
            parts[k].append((b.mass, p + R @ b.ipos, R @ b.inertia @ R.T))
        fb_inertia = {k: np.zeros((3, 3)) for k in keep}
        for k in keep:
            m = sum(x[0] for x in parts[k])
            if m > 0:
                com = sum(x[0] * x[1] for x in parts[k]) / m
            else:
                com = np.zeros(3)
            I = np.zeros((3, 3))
            for pm, pc, pI in parts[k]:
                d = pc - com
                I += pI + pm * (np.dot(d, d) * np.eye(3) - np.outer(d, d))
            fb_mass[k], fb_com[k], fb_inertia[k] = m, com, I

        T: Dict[str, np.ndarray] = {}
        names: Dict[str, Dict[str, int]] = {k: {} for k in ("body", "joint", "geom", "site", "actuator", "mocap", "body_orig")}
        info: Dict[str, object] = {"xml": os.path.basename(c.xml_path)}
        nbk = len(keep)
        body_parent = np.zeros(nbk, np.int32)
        body_depth = np.zeros(nbk, np.int32)
        body_rootid = np.zeros(nbk, np.int32)
        for k, old in enumerate(keep):
            names["body"][B[old].name] = k
            if k == 0:
                continue
            par, _, _ = anchor_of(B[old].parent)
            body_parent[k] = new_id[par]
            body_depth[k] = body_depth[body_parent[k]] + 1
            body_rootid[k] = k if body_parent[k] == 0 else body_rootid[body_parent[k]]
        # original body name -> (fused id, rel pos, rel quat)
        body_orig = {}
        for i in range(nb):
            k, p, q = anchor_of(i)
            body_orig[B[i].name] = (new_id[k], p.tolist(), q.tolist())
        info["body_orig"] = body_orig
        # per-world shift group (c.shift_body): the named static child of the world and every body welded to it
        shift_set, shift_jointed = set(), -1
        if c.shift_body is not None:
            root_i = next(i for i, b in enumerate(B) if b.name == c.shift_body)
            if B[root_i].parent != 0:
                raise NotImplementedError("shift_body must be a child of the world")
            if B[root_i].joints:
                if c.shift_rotates:
                    raise NotImplementedError("a rotating shift group must be static")
                shift_jointed = root_i
            else:
                shift_set = {root_i}
                grew = True
                while grew:
                    grew = False
                    for i in range(1, nb):
                        if i not in shift_set and B[i].parent in shift_set and not B[i].joints and not B[i].mocap:
                            shift_set.add(i); grew = True
                if c.shift_rotates and any(B[i].parent in shift_set and i not in shift_set for i in range(1, nb)):
                    raise NotImplementedError("a rotating shift group must not carry moving children")
            info["shift_pos0"] = B[root_i].pos.tolist()
            info["shift_quat0"] = B[root_i].quat.tolist()
        shift_flag = 2 if c.shift_rotates else 1

        body_shift = np.zeros(nbk, np.int32)
        body_pos = np.zeros((nbk, 3))
        body_quat = np.tile(np.array([1.0, 0, 0, 0]), (nbk, 1))
        body_ipos = np.zeros((nbk, 3))
        body_inertia = np.zeros((nbk, 6))
        body_mass = np.zeros(nbk)
        body_jntadr = -np.ones(nbk, np.int32)
        body_jntnum = np.zeros(nbk, np.int32)
        body_dofadr = -np.ones(nbk, np.int32)
        body_dofnum = np.zeros(nbk, np.int32)
        body_mocapid = -np.ones(nbk, np.int32)
        nmocap = 0
        mocap_pos0, mocap_quat0 = [], []
        jcount = 0
        for k, old in enumerate(keep):
            if k == 0:
                continue
            b = B[old]
            # pose relative to the kept parent: compose through fused static ancestors
            par_old = b.parent
            pk, pp, pq = anchor_of(par_old)
            body_pos[k] = pp + mu.rot_vec(pq, b.pos)
            body_shift[k] = int(par_old in shift_set or old == shift_jointed)   # a moving child of the shift group: its world-frame origin moves with the group
            body_quat[k] = mu.quat_normalize(mu.quat_mul(pq, b.quat))
            body_ipos[k] = fb_com[old]
            I = fb_inertia[old]
            body_inertia[k] = [I[0, 0], I[1, 1], I[2, 2], I[0, 1], I[0, 2], I[1, 2]]
            body_mass[k] = fb_mass[old]
            if b.joints:
                body_jntadr[k] = jcount
                body_jntnum[k] = len(b.joints)
                body_dofadr[k] = b.joints[0].dofadr
                body_dofnum[k] = sum({JNT_FREE: 6, JNT_BALL: 3}.get(j.type, 1) for j in b.joints)
                jcount += len(b.joints)
            if b.mocap:
                body_mocapid[k] = nmocap
                names["mocap"][b.name] = nmocap
                nmocap += 1
                mocap_pos0.append(body_pos[k].copy())
                mocap_quat0.append(body_quat[k].copy())

        njnt = len(jnts)
        qpos0 = np.zeros(nq)
        for j in jnts:
            names["joint"][j.name] = jnts.index(j)
            if j.type == JNT_FREE:
                bi = j.body
                qpos0[j.qposadr: j.qposadr + 3] = xpos[bi]
                qpos0[j.qposadr + 3: j.qposadr + 7] = xquat[bi]
            elif j.type == JNT_BALL:
                qpos0[j.qposadr: j.qposadr + 4] = [1, 0, 0, 0]
            else:
                qpos0[j.qposadr] = j.ref

        def arr(fn, dtype=np.float64):
            return np.array([fn(j) for j in jnts], dtype=dtype).reshape(njnt, -1) if njnt else np.zeros((0, 1), dtype)

        T["jnt_type"] = arr(lambda j: j.type, np.int32)
        T["jnt_qposadr"] = arr(lambda j: j.qposadr, np.int32)
        T["jnt_dofadr"] = arr(lambda j: j.dofadr, np.int32)
        T["jnt_bodyid"] = arr(lambda j: new_id[j.body], np.int32)
        T["jnt_limited"] = arr(lambda j: int(j.limited), np.int32)
        T["jnt_pos"] = arr(lambda j: j.pos)
        T["jnt_axis"] = arr(lambda j: j.axis)
        T["jnt_range"] = arr(lambda j: j.range)
        T["jnt_margin"] = arr(lambda j: j.margin)
        T["jnt_solref"] = arr(lambda j: j.solref)
        T["jnt_solimp"] = arr(lambda j: j.solimp)
        T["jnt_stiffness"] = arr(lambda j: j.stiffness)
        # spring rest value: for slide/hinge q_spring = springref (MJCF), offset handled by engine as q - springref
        T["jnt_springref"] = arr(lambda j: j.springref)

        dof_bodyid = np.zeros(nv, np.int32)
        dof_jntid = np.zeros(nv, np.int32)
        dof_parentid = -np.ones(nv, np.int32)
        dof_arm = np.zeros(nv)
        dof_damp = np.zeros(nv)
        dof_fl = np.zeros(nv)
        dof_solref = np.zeros((nv, 2))
        dof_solimp = np.zeros((nv, 5))
        last_dof_of_body = {0: -1}
        for k, old in enumerate(keep):
            if k == 0:
                continue
            prev = last_dof_of_body[int(body_parent[k])]
            for j in B[old].joints:
                n = {JNT_FREE: 6, JNT_BALL: 3}.get(j.type, 1)
                for t in range(n):
                    d = j.dofadr + t
                    dof_bodyid[d] = k
                    dof_jntid[d] = jnts.index(j)
                    dof_parentid[d] = prev
                    dof_arm[d] = j.armature
                    dof_damp[d] = j.damping
                    dof_fl[d] = j.frictionloss
                    dof_solref[d] = j.solreffriction
                    dof_solimp[d] = j.solimpfriction
                    prev = d
            last_dof_of_body[k] = prev

        # ---- geoms (collidable only) and sites, re-expressed in fused frames
        geoms = []
        for i in range(nb):
            for g in B[i].geoms:
                if g.contype == 0 and g.conaffinity == 0:
                    continue
                geoms.append((i, g))
        ng = len(geoms)
        geom_type = np.zeros(ng, np.int32)
        geom_bodyid = np.zeros(ng, np.int32)
        geom_meshadr = -np.ones(ng, np.int32)
        geom_meshnum = np.zeros(ng, np.int32)
        geom_hulladr = -np.ones(ng, np.int32)   # full convex hull (support function of the hull-vs-convex pairs); geom_meshadr may be a pruned subset for the plane pairs
        geom_hullnum = np.zeros(ng, np.int32)
        geom_pos = np.zeros((ng, 3))
        geom_quat = np.zeros((ng, 4))
        geom_size = np.zeros((ng, 3))
        geom_invw = np.zeros((ng, 2))
        geom_shift = np.zeros(ng, np.int32)
        geom_rbound = np.zeros(ng)
        geom_aabb = np.zeros((ng, 6))   # centre, half extents in the geom frame
        mesh_vert, mesh_adjadr, mesh_adjnum, mesh_adj = [], [], [], []
        mesh_cache = {}
        for gi, (i, g) in enumerate(geoms):
            k, p, q = anchor_of(i)
            names["geom"][g.name or f"_geom{gi}"] = gi
            geom_type[gi] = g.type
            geom_bodyid[gi] = new_id[k]
            geom_shift[gi] = shift_flag * int(i in shift_set)
            geom_pos[gi] = p + mu.rot_vec(q, g.pos)
            geom_quat[gi] = mu.quat_normalize(mu.quat_mul(q, g.quat))
            geom_size[gi] = g.size
            geom_invw[gi] = body_invweight0[i]
            s = g.size
            if g.type == GEOM_MESH:
                # The geom frame of a mesh is moved to the centre of mass of its hull (MuJoCo re-centres a mesh at its centre of mass as
                # well; it also rotates the frame to the principal axes, which no routine here depends on): the convex routine casts its
                # first ray from the geom centre, which has to be an interior point.  World geometry is unchanged.
                hv, _ = c._load_mesh_hull(g.mesh)
                ctr = c._mesh_center(g.mesh)
                geom_pos[gi] = geom_pos[gi] + mu.rot_vec(geom_quat[gi], ctr)
                hvc = hv - ctr
                geom_rbound[gi] = np.linalg.norm(hvc, axis=1).max()
                lo, hi = hvc.min(axis=0), hvc.max(axis=0)
                geom_aabb[gi] = np.concatenate([0.5 * (lo + hi), 0.5 * (hi - lo)])
            elif g.type == GEOM_SPHERE:
                geom_rbound[gi] = s[0]
            elif g.type == GEOM_CAPSULE:
                geom_rbound[gi] = s[0] + s[1]
            elif g.type == GEOM_CYLINDER:
                geom_rbound[gi] = np.hypot(s[0], s[1])
            elif g.type in (GEOM_BOX, GEOM_ELLIPSOID):
                geom_rbound[gi] = np.linalg.norm(s) if g.type == GEOM_BOX else s.max()
            if g.type == GEOM_SPHERE:
                geom_aabb[gi, 3:] = s[0]
            elif g.type == GEOM_CAPSULE:
                geom_aabb[gi, 3:] = [s[0], s[0], s[0] + s[1]]
            elif g.type == GEOM_CYLINDER:
                geom_aabb[gi, 3:] = [s[0], s[0], s[1]]
            elif g.type in (GEOM_BOX, GEOM_ELLIPSOID):
                geom_aabb[gi, 3:] = s
        # touch sensors selected by touch_filter: their zones (sites) go to the touch_* tables, in sensor order, and are left out of
        # the engine's site tables (92 zones x 12 words of world frames per world would not pay for themselves in LDS)
        touch_sel = []
        if c.touch_filter is not None:
            for sec in c.root.findall("sensor"):
                for e in sec.findall("touch"):
                    if c.touch_filter(e.attrib.get("name", "")):
                        touch_sel.append((e.attrib.get("name", ""), e.attrib["site"]))
        touch_sites = {sn for _, sn in touch_sel}
        all_sites = {s.name: (i, s) for i in range(nb) for s in B[i].sites if s.name}
        nt = len(touch_sel)
        touch_body, touch_type = np.zeros(nt, np.int32), np.zeros(nt, np.int32)
        touch_pos, touch_quat, touch_size = np.zeros((nt, 3)), np.zeros((nt, 4)), np.zeros((nt, 3))
        for ti, (sname, site_name) in enumerate(touch_sel):
            i, sdef = all_sites[site_name]
            if sdef.type not in (GEOM_SPHERE, GEOM_BOX, GEOM_CYLINDER):
                raise NotImplementedError("touch zones: sphere, box and cylinder sites only")
            k, p, q = anchor_of(i)
            touch_body[ti], touch_type[ti] = new_id[k], sdef.type
            touch_pos[ti] = p + mu.rot_vec(q, sdef.pos)
            touch_quat[ti] = mu.quat_normalize(mu.quat_mul(q, sdef.quat))
            touch_size[ti] = sdef.size
        T.update(touch_body=touch_body, touch_type=touch_type, touch_pos=touch_pos, touch_quat=touch_quat, touch_size=touch_size)
        names["touch"] = {sname: ti for ti, (sname, _) in enumerate(touch_sel)}
        sites = [(i, s) for i in range(nb) for s in B[i].sites
                 if (s.name not in touch_sites and c.keep_sites is None) or (c.keep_sites is not None and s.name in c.keep_sites)]   # a touch zone stays an engine site only when asked for by name
        ns = len(sites)
        site_bodyid = np.zeros(ns, np.int32)
        site_shift = np.zeros(ns, np.int32)
        site_type = np.zeros(ns, np.int32)
        site_pos = np.zeros((ns, 3))
        site_quat = np.zeros((ns, 4))
        site_size = np.zeros((ns, 3))
        for si, (i, s) in enumerate(sites):
            k, p, q = anchor_of(i)
            names["site"][s.name or f"_site{si}"] = si
            site_bodyid[si] = new_id[k]
            site_shift[si] = shift_flag * int(i in shift_set)
            site_type[si] = s.type
            site_pos[si] = p + mu.rot_vec(q, s.pos)
            site_quat[si] = mu.quat_normalize(mu.quat_mul(q, s.quat))
            site_size[si] = s.size

        # ---- collision candidate pairs
        excludes = set()
        for ct in c.root.findall("contact"):
            for ex in ct.findall("exclude"):
                b1 = next(i for i, b in enumerate(B) if b.name == ex.attrib["body1"])
                b2 = next(i for i, b in enumerate(B) if b.name == ex.attrib["body2"])
                excludes.add((min(b1, b2), max(b1, b2)))
        weldparent = [weld[B[weld[i]].parent] if weld[i] > 0 else 0 for i in range(nb)]
        # narrow-phase routines implemented by BOTH the device engine and the oracle
        supported = {(GEOM_PLANE, GEOM_BOX), (GEOM_PLANE, GEOM_MESH), (GEOM_BOX, GEOM_BOX), (GEOM_PLANE, GEOM_SPHERE), (GEOM_SPHERE, GEOM_BOX),
                     (GEOM_PLANE, GEOM_CAPSULE), (GEOM_CAPSULE, GEOM_BOX), (GEOM_CAPSULE, GEOM_CAPSULE), (GEOM_PLANE, GEOM_ELLIPSOID), (GEOM_PLANE, GEOM_CYLINDER)}
        # general convex narrow phase (MPR, one contact): every pair of primitives that involves an ellipsoid or a cylinder
        prim = (GEOM_SPHERE, GEOM_CAPSULE, GEOM_ELLIPSOID, GEOM_CYLINDER, GEOM_BOX)
        supported |= {(ta, tb) for ta in prim for tb in prim if ta <= tb and (GEOM_ELLIPSOID in (ta, tb) or GEOM_CYLINDER in (ta, tb))}
        supported |= {(GEOM_SPHERE, GEOM_SPHERE), (GEOM_SPHERE, GEOM_CAPSULE)}
        # convex hull of a mesh against any primitive or another hull: the same routine with a hull-vertex support function
        supported |= {(ta, GEOM_MESH) for ta in prim + (GEOM_MESH,)}
        pairs = []
        for a in range(ng):
            for b_ in range(a + 1, ng):
                (i1, g1), (i2, g2) = geoms[a], geoms[b_]
                w1, w2 = weld[i1], weld[i2]
                if w1 == w2:
                    continue
                if w1 != 0 and w2 != 0 and (w1 == weldparent[i2] or w2 == weldparent[i1]):
                    continue
                if not ((g1.contype & g2.conaffinity) or (g2.contype & g1.conaffinity)):
                    continue
                if (min(i1, i2), max(i1, i2)) in excludes:
                    continue
                ga, gb, A_, B_ = (g1, g2, a, b_) if g1.type <= g2.type else (g2, g1, b_, a)
                if ga.type == GEOM_PLANE and gb.type == GEOM_PLANE:
                    continue
                pairs.append((A_, B_, ga, gb))
        # explicit <contact><pair> elements (MuJoCo XML reference, contact/pair): they bypass the contype/conaffinity and
        # parent-child filters, take their parameters from the pair defaults (condim 3, friction 1 1 0.005 0.0001 0.0001,
        # solref 0.02 1, solimp 0.9 0.95 0.001 0.5 2, margin 0, gap 0) unless given, and replace a dynamically generated pair
        # of the same two geoms.  A pair listed twice is kept once.
        gindex = {g.name: gi for gi, (i, g) in enumerate(geoms) if g.name}
        explicit = {}
        for ct in c.root.findall("contact"):
            for pe in ct.findall("pair"):
                a = dict(c.defaults.get(pe.attrib.get("class"), "pair"))
                a.update(pe.attrib)
                if a["geom1"] not in gindex or a["geom2"] not in gindex:
                    raise ValueError(f"contact pair refers to unknown geom {a['geom1']} / {a['geom2']}")
                ia, ib = gindex[a["geom1"]], gindex[a["geom2"]]
                ga, gb = geoms[ia][1], geoms[ib][1]
                if ga.type > gb.type:
                    ia, ib, ga, gb = ib, ia, gb, ga
                key = (min(ia, ib), max(ia, ib))
                if key in explicit:
                    continue
                fr = _floats(a.get("friction"), 5, [1, 1, 0.005, 0.0001, 0.0001])
                explicit[key] = dict(A=ia, B=ib, ga=ga, gb=gb, condim=int(a.get("condim", 3)), friction=fr,
                                     solref=_floats(a.get("solref"), 2, [0.02, 1.0]), solimp=_floats(a.get("solimp"), 5, [0.9, 0.95, 0.001, 0.5, 2]),
                                     margin=float(a.get("margin", 0.0)), gap=float(a.get("gap", 0.0)))
        pairs = [pr for pr in pairs if (min(pr[0], pr[1]), max(pr[0], pr[1])) not in explicit]
        ndyn = len(pairs)
        for key in sorted(explicit):
            e = explicit[key]
            pairs.append((e["A"], e["B"], e["ga"], e["gb"]))
        npair = len(pairs)
        pair_geom1 = np.zeros(npair, np.int32)
        pair_geom2 = np.zeros(npair, np.int32)
        pair_condim = np.zeros(npair, np.int32)
        pair_supported = np.zeros(npair, np.int32)
        pair_friction = np.zeros((npair, 5))
        pair_solref = np.zeros((npair, 2))
        pair_solimp = np.zeros((npair, 5))
        pair_margin = np.zeros(npair)
        pair_gap = np.zeros(npair)
        for pi, (A_, B_, ga, gb) in enumerate(pairs):
            pair_geom1[pi], pair_geom2[pi] = A_, B_
            pair_supported[pi] = int((ga.type, gb.type) in supported)
            if pi >= ndyn:
                e = explicit[(min(A_, B_), max(A_, B_))]
                pair_condim[pi] = e["condim"]
                pair_friction[pi] = e["friction"]
                pair_solref[pi], pair_solimp[pi] = e["solref"], e["solimp"]
                pair_margin[pi], pair_gap[pi] = e["margin"], e["gap"]
                continue
            if ga.priority != gb.priority:
                gp = ga if ga.priority > gb.priority else gb
                pair_condim[pi] = gp.condim
                fr = gp.friction
                pair_solref[pi], pair_solimp[pi] = gp.solref, gp.solimp
            else:
                pair_condim[pi] = max(ga.condim, gb.condim)
                fr = np.maximum(ga.friction, gb.friction)
                if ga.solmix >= MJ_MINVAL and gb.solmix >= MJ_MINVAL:
                    mix = ga.solmix / (ga.solmix + gb.solmix)
                elif ga.solmix < MJ_MINVAL and gb.solmix < MJ_MINVAL:
                    mix = 0.5
                elif ga.solmix < MJ_MINVAL:
                    mix = 0.0
                else:
                    mix = 1.0
                if ga.solref[0] > 0 and gb.solref[0] > 0:
                    pair_solref[pi] = mix * ga.solref + (1 - mix) * gb.solref
                else:
                    pair_solref[pi] = np.minimum(ga.solref, gb.solref)
                pair_solimp[pi] = mix * ga.solimp + (1 - mix) * gb.solimp
            pair_friction[pi] = [fr[0], fr[0], fr[1], fr[2], fr[2]]
            pair_margin[pi] = max(ga.margin, gb.margin)
            pair_gap[pi] = max(ga.gap, gb.gap)

        # ---- hull vertex tables.  A mesh whose body can only translate (slide joints all the way up) keeps a constant
        # orientation, so against a static plane the deepest hull vertex and its hull neighbours are known at compile
        # time: keep only those (the plane-mesh routine uses exactly "deepest vertex + its hull neighbours within margin").
        def only_slides(i):
            while i > 0:
                if any(j.type != JNT_SLIDE for j in B[i].joints):
                    return False
                i = B[i].parent
            return True

        def emit_hull(hv2, adj2):
            adr = len(mesh_vert)
            for v, a_ in zip(hv2, adj2):
                mesh_vert.append(v)
                mesh_adjadr.append(len(mesh_adj))
                mesh_adjnum.append(len(a_))
                mesh_adj.extend(a_)
            return adr, len(hv2)

        for gi, (i, g) in enumerate(geoms):
            if g.type != GEOM_MESH:
                continue
            my_pairs = [(A_, B_, ga, gb) for (A_, B_, ga, gb) in pairs if gi in (A_, B_)]
            sup = [pr for pr in my_pairs if (pr[2].type, pr[3].type) in supported]
            if not sup:
                continue  # no implemented narrow phase touches this mesh: no vertices needed on the device
            hv, adj = c._load_mesh_hull(g.mesh)
            hv = hv - c._mesh_center(g.mesh)   # geom frame = hull centre of mass (see geom_pos above)
            psup = [pr for pr in sup if pr[2].type == GEOM_PLANE]          # plane pairs: deepest vertex + neighbours (geom_meshadr / geom_meshnum)
            csup = [pr for pr in sup if pr[2].type != GEOM_PLANE]          # hull-vs-convex pairs: support function over the FULL hull (geom_hulladr / geom_hullnum)
            full = None
            if psup:
                keep_v = None
                planes_static = all(weld[geoms[pr[0]][0]] == 0 for pr in psup)
                if planes_static and only_slides(i):
                    Rg = xmat[i] @ mu.quat2mat(g.quat)
                    keep_set = set()
                    for pr in psup:
                        pi_, pg = geoms[pr[0]]
                        n_w = (xmat[pi_] @ mu.quat2mat(pg.quat))[:, 2]
                        h = (hv @ Rg.T) @ n_w
                        vstar = int(np.argmin(h))
                        keep_set |= {vstar} | set(adj[vstar])
                    keep_v = sorted(keep_set)
                if keep_v is not None:
                    remap = {v: k for k, v in enumerate(keep_v)}
                    geom_meshadr[gi], geom_meshnum[gi] = emit_hull(hv[keep_v], [[remap[w] for w in adj[v] if w in remap] for v in keep_v])
                else:
                    full = emit_hull(hv, adj)
                    geom_meshadr[gi], geom_meshnum[gi] = full
            if csup:
                if full is None:
                    full = emit_hull(hv, adj)
                geom_hulladr[gi], geom_hullnum[gi] = full
                if not psup:
                    geom_meshadr[gi], geom_meshnum[gi] = full

        # ---- equality constraints
        eqs = []
        for eqsec in c.root.findall("equality"):
            for e in eqsec:
                a = dict(c.defaults.get(e.attrib.get("class"), "equality"))
                a.update(e.attrib)
                if e.tag == "joint":
                    # q1 - q1_0 = poly(q2 - q2_0) (MuJoCo XML reference, equality/joint; kitchen_franka/.../oven_asset.xml:40-46: knob <-> burner).
                    # One row, J = e_dof1 - poly'(.) e_dof2; diagApprox = dof_invweight0[dof1] + dof_invweight0[dof2] (mj_setConst [3P]).
                    j1 = jnts[names["joint"][a["joint1"]]]
                    if "joint2" not in a:
                        raise NotImplementedError("joint equality without joint2")
                    j2 = jnts[names["joint"][a["joint2"]]]
                    if j1.type not in (JNT_HINGE, JNT_SLIDE) or j2.type not in (JNT_HINGE, JNT_SLIDE):
                        raise ValueError("joint equalities couple hinge / slide joints only")
                    data = np.zeros(11)
                    data[0:5] = _floats(a.get("polycoef"), 5, [0, 1, 0, 0, 0])
                    data[5], data[6] = qpos0[j1.qposadr], qpos0[j2.qposadr]
                    iw = float(dof_invweight0[j1.dofadr] + dof_invweight0[j2.dofadr])
                    eqs.append(dict(
                        type=EQ_JOINT, obj1=int(names["joint"][a["joint1"]]), obj2=int(names["joint"][a["joint2"]]), active=int(_bool(a.get("active"), True)),
                        data=data, solref=_floats(a.get("solref"), 2, [0.02, 1.0]), solimp=_floats(a.get("solimp"), 5, [0.9, 0.95, 0.001, 0.5, 2.0]),
                        invw=np.array([iw, iw]), relpose=np.zeros(14), dofs=(int(j1.dofadr), int(j2.dofadr)), qadrs=(int(j1.qposadr), int(j2.qposadr)),
                    ))
                    continue
                if e.tag != "weld":
                    raise NotImplementedError(f"equality type {e.tag}")
                i1 = next(i for i, b in enumerate(B) if b.name == a["body1"])
                i2 = next(i for i, b in enumerate(B) if b.name == a.get("body2", "world"))
                k1, p1, q1 = anchor_of(i1)
                k2, p2, q2 = anchor_of(i2)
                anchor = _floats(a.get("anchor"), 3, [0, 0, 0])
                R1 = xmat[i1]
                relpos = R1.T @ (xpos[i2] + xmat[i2] @ anchor - xpos[i1])
                relquat = mu.quat_mul(mu.quat_conj(xquat[i1]), xquat[i2])
                if "relpose" in a:
                    rp = _floats(a["relpose"])
                    if np.any(rp[3:] != 0) and not np.allclose(rp, [0, 1, 0, 0, 0, 0, 0]):
                        relpos, relquat = rp[:3], mu.quat_normalize(rp[3:])
                data = np.zeros(11)
                data[0:3] = anchor
                data[3:6] = relpos
                data[6:10] = relquat
                data[10] = float(a.get("torquescale", 1.0))
                eqs.append(dict(
                    type=EQ_WELD, obj1=new_id[k1], obj2=new_id[k2], active=int(_bool(a.get("active"), True)),
                    data=data, solref=_floats(a.get("solref"), 2, [0.02, 1.0]),
                    solimp=_floats(a.get("solimp"), 5, [0.9, 0.95, 0.001, 0.5, 2.0]),
                    invw=body_invweight0[i1] + body_invweight0[i2],
                        relpose=np.concatenate([p1, q1, p2, q2]),
                ))
        neq = len(eqs)

        # ---- actuators
        acts = []
        for asec in c.root.findall("actuator"):
            for e in asec:
                cls = e.attrib.get("class")
                a = dict(c.defaults.get(cls, "general"))
                own = _actuator_shortcut(e.tag, dict(e.attrib)) if e.tag != "general" else dict(e.attrib)
                a.update(own)
                if a.get("_position") or e.tag == "position":
                    kp = float(a.get("_kp", _floats(a.get("gainprm"), 3, [1, 0, 0])[0]))
                    kv = float(a.get("_kv", 0.0))
                    a["gainprm"] = f"{kp} 0 0"
                    a["biasprm"] = f"0 {-kp} {-kv}"
                    a["biastype"] = "affine"
                    a["gaintype"] = "fixed"
                if "joint" not in a:
                    raise NotImplementedError("only joint transmissions are supported")
                jid = names["joint"][a["joint"]]
                ctrlrange = _floats(a.get("ctrlrange"), 2, [0, 0])
                forcerange = _floats(a.get("forcerange"), 2, [0, 0])
                cl = a.get("ctrllimited", "auto")
                fl = a.get("forcelimited", "auto")
                acts.append(dict(
                    name=a.get("name", f"_act{len(acts)}"), trnid=jid,
                    gaintype={"fixed": 0, "affine": 1}[a.get("gaintype", "fixed")],
                    biastype={"none": 0, "affine": 1}[a.get("biastype", "none")],
                    ctrllimited=int(_bool(cl) if cl != "auto" else (c.autolimits and "ctrlrange" in a)),
                    forcelimited=int(_bool(fl) if fl != "auto" else (c.autolimits and "forcerange" in a)),
                    gear=_floats(a.get("gear"), 6, [1, 0, 0, 0, 0, 0])[0],
                    gainprm=_floats(a.get("gainprm"), 3, [1, 0, 0])[:3],
                    biasprm=_floats(a.get("biasprm"), 3, [0, 0, 0])[:3],
                    ctrlrange=ctrlrange, forcerange=forcerange,
                ))
        nu = len(acts)
        for i, a in enumerate(acts):
            names["actuator"][a["name"]] = i

        # ---- fixed tendons.  Only their limit rows are on the path (the models in scope attach no actuator, spring, damper
        # or friction loss to a tendon): length = sum coef * qpos, Jacobian = the coefficients, invweight0 = J M^-1 J' at qpos0.
        tendons, wrap_dof, wrap_qadr, wrap_coef = [], [], [], []
        for tsec in c.root.findall("tendon"):
            for e in tsec:
                if e.tag != "fixed":
                    raise NotImplementedError("only fixed tendons are supported")
                a = dict(c.defaults.get(e.attrib.get("class"), "tendon"))
                a.update(e.attrib)
                if any(float(a.get(k, 0.0)) != 0.0 for k in ("stiffness", "damping", "frictionloss")):
                    raise NotImplementedError("tendon stiffness / damping / frictionloss")
                Jt = np.zeros(nv)
                adr = len(wrap_dof)
                for w in e.findall("joint"):
                    j = jnts[names["joint"][w.attrib["joint"]]]
                    if j.type not in (JNT_HINGE, JNT_SLIDE):
                        raise ValueError("fixed tendons couple hinge / slide joints only")
                    wrap_dof.append(j.dofadr); wrap_qadr.append(j.qposadr); wrap_coef.append(float(w.attrib["coef"]))
                    Jt[j.dofadr] += float(w.attrib["coef"])
                nz = np.nonzero(Jt)[0]
                lo = int(nz.min()) if len(nz) else 0
                ln = int(nz.max()) - lo + 1 if len(nz) else 0
                lim = a.get("limited", "auto")
                tendons.append(dict(
                    adr=adr, num=len(wrap_dof) - adr, span=lo | (ln << 8),
                    limited=int(_bool(lim) if lim != "auto" else (c.autolimits and "range" in a)),
                    range=_floats(a.get("range"), 2, [0, 0]), margin=float(a.get("margin", 0.0)),
                    solref=_floats(a.get("solreflimit"), 2, [0.02, 1.0]), solimp=_floats(a.get("solimplimit"), 5, [0.9, 0.95, 0.001, 0.5, 2]),
                    invweight0=float(Jt @ Minv @ Jt) if nv else 0.0,
                ))
        ntendon = len(tendons)
        if ntendon > 64:
            raise ValueError("engine limit: at most 64 tendons (one lane per tendon)")
        T.update(
            tendon_adr=np.array([t["adr"] for t in tendons], np.int32), tendon_num=np.array([t["num"] for t in tendons], np.int32),
            tendon_limited=np.array([t["limited"] for t in tendons], np.int32), tendon_span=np.array([t["span"] for t in tendons], np.int32),
            tendon_range=np.array([t["range"] for t in tendons]).reshape(ntendon, 2), tendon_margin=np.array([t["margin"] for t in tendons], np.float64),
            tendon_solref=np.array([t["solref"] for t in tendons]).reshape(ntendon, 2),
            tendon_solimp=np.array([t["solimp"] for t in tendons]).reshape(ntendon, 5),
            tendon_invweight0=np.array([t["invweight0"] for t in tendons], np.float64),
            wrap_dof=np.array(wrap_dof, np.int32), wrap_qadr=np.array(wrap_qadr, np.int32), wrap_coef=np.array(wrap_coef, np.float64),
        )

        # ---- assemble tables
        dims = np.zeros(NDIMS, np.int32)
        vals = dict(nq=nq, nv=nv, nu=nu, nbody=nbk, njnt=njnt, ngeom=ng, nsite=ns, nmocap=nmocap, neq=neq,
                    npair=npair, nmeshvert=len(mesh_vert), nmeshadj=len(mesh_adj),
                    integrator=c.opt["integrator"], iterations=c.opt["iterations"], cone=c.opt["cone"],
                    noslip_iterations=c.opt["noslip_iterations"], eulerdamp=c.opt["eulerdamp"],
                    ntree=int(np.sum(body_parent[1:] == 0)), maxdepth=int(body_depth.max()),
                    maxefc_req=int(c.capacity.get("maxefc", 0)), jpool_req=int(c.capacity.get("jpool", 0)),
                    maxcon_req=int(c.capacity.get("maxcon", 0)))
        for k, v in vals.items():
            dims[DIMS.index(k)] = v
        optv = np.zeros(NOPTS)
        g = c.opt["gravity"]
        for k, v in dict(timestep=c.opt["timestep"], gravity_x=g[0], gravity_y=g[1], gravity_z=g[2],
                         tolerance=c.opt["tolerance"], impratio=c.opt["impratio"], meaninertia=meaninertia,
                         mpr_tolerance=c.opt["mpr_tolerance"], mpr_iterations=c.opt["mpr_iterations"],
                         noslip_tolerance=c.opt["noslip_tolerance"], origin_x=c.origin[0], origin_y=c.origin[1], origin_z=c.origin[2]).items():
            optv[OPTS.index(k)] = v
        T.update(
            dims=dims, opt=optv, qpos0=qpos0,
            body_shift=body_shift, geom_shift=geom_shift, site_shift=site_shift,
            body_parent=body_parent, body_jntadr=body_jntadr, body_jntnum=body_jntnum, body_dofadr=body_dofadr,
            body_dofnum=body_dofnum, body_mocapid=body_mocapid, body_rootid=body_rootid, body_depth=body_depth,
            body_pos=body_pos, body_quat=body_quat, body_ipos=body_ipos, body_inertia=body_inertia, body_mass=body_mass,
            dof_bodyid=dof_bodyid, dof_jntid=dof_jntid, dof_parentid=dof_parentid, dof_armature=dof_arm,
            dof_damping=dof_damp, dof_frictionloss=dof_fl, dof_invweight0=dof_invweight0,
            dof_solref=dof_solref, dof_solimp=dof_solimp,
            geom_type=geom_type, geom_bodyid=geom_bodyid, geom_meshadr=geom_meshadr, geom_meshnum=geom_meshnum,
            geom_pos=geom_pos, geom_quat=geom_quat, geom_size=geom_size, geom_invweight0=geom_invw, geom_rbound=geom_rbound, geom_aabb=geom_aabb, geom_hulladr=geom_hulladr, geom_hullnum=geom_hullnum,
            site_bodyid=site_bodyid, site_type=site_type, site_pos=site_pos, site_quat=site_quat, site_size=site_size,
            pair_geom1=pair_geom1, pair_geom2=pair_geom2, pair_condim=pair_condim, pair_supported=pair_supported,
            pair_friction=pair_friction, pair_solref=pair_solref, pair_solimp=pair_solimp, pair_margin=pair_margin,
            pair_gap=pair_gap,
            eq_type=np.array([e["type"] for e in eqs], np.int32), eq_obj1=np.array([e["obj1"] for e in eqs], np.int32),
            eq_obj2=np.array([e["obj2"] for e in eqs], np.int32), eq_active=np.array([e["active"] for e in eqs], np.int32),
            eq_data=np.array([e["data"] for e in eqs]).reshape(neq, 11), eq_solref=np.array([e["solref"] for e in eqs]).reshape(neq, 2),
            eq_solimp=np.array([e["solimp"] for e in eqs]).reshape(neq, 5), eq_invweight=np.array([e["invw"] for e in eqs]).reshape(neq, 2),
            eq_relpose=np.array([e["relpose"] for e in eqs]).reshape(neq, 14),
            act_trntype=np.zeros(nu, np.int32), act_trnid=np.array([a["trnid"] for a in acts], np.int32),
            act_gaintype=np.array([a["gaintype"] for a in acts], np.int32), act_biastype=np.array([a["biastype"] for a in acts], np.int32),
            act_ctrllimited=np.array([a["ctrllimited"] for a in acts], np.int32),
            act_forcelimited=np.array([a["forcelimited"] for a in acts], np.int32),
            act_gear=np.array([a["gear"] for a in acts], np.float64),
            act_gainprm=np.array([a["gainprm"] for a in acts]).reshape(nu, 3), act_biasprm=np.array([a["biasprm"] for a in acts]).reshape(nu, 3),
            act_ctrlrange=np.array([a["ctrlrange"] for a in acts]).reshape(nu, 2),
            act_forcerange=np.array([a["forcerange"] for a in acts]).reshape(nu, 2),
            mocap_pos0=np.array(mocap_pos0).reshape(nmocap, 3), mocap_quat0=np.array(mocap_quat0).reshape(nmocap, 4),
            mesh_vert=np.array(mesh_vert).reshape(len(mesh_vert), 3), mesh_adjadr=np.array(mesh_adjadr, np.int32),
            mesh_adjnum=np.array(mesh_adjnum, np.int32), mesh_adj=np.array(mesh_adj, np.int32),
        )
        # ---- derived tree tables (static structure used by the wave-per-world engine)
        order = sorted(range(1, nbk), key=lambda k: (body_depth[k], k))
        maxdepth = int(body_depth.max()) if nbk > 1 else 0
        level_adr = np.zeros(maxdepth + 2, np.int32)
        for k in order:
            level_adr[body_depth[k] + 1] += 1
        level_adr = np.cumsum(level_adr).astype(np.int32)
        subs = [[k] for k in range(nbk)]
        for k in range(nbk - 1, 0, -1):
            if body_parent[k] > 0:
                subs[body_parent[k]] = subs[body_parent[k]] + subs[k]
        subs[0] = [0]
        body_subadr = np.zeros(nbk, np.int32)
        body_subnum = np.zeros(nbk, np.int32)
        body_sub = []
        for k in range(nbk):
            body_subadr[k], body_subnum[k] = len(body_sub), len(subs[k])
            body_sub += sorted(subs[k])
        body_lastdof = -np.ones(nbk, np.int32)
        for k in range(1, nbk):
            if body_dofnum[k] > 0:
                body_lastdof[k] = body_dofadr[k] + body_dofnum[k] - 1
            else:
                body_lastdof[k] = body_lastdof[body_parent[k]]
        mpi, mpj = [], []
        for i in range(nv):
            j = i
            while j >= 0:
                mpi.append(i)
                mpj.append(j)
                j = int(dof_parentid[j])
        dof_cvelstart = np.zeros(nv, np.int32)
        for d in range(nv):
            j = jnts[dof_jntid[d]]
            if j.type == JNT_FREE:
                t = d - j.dofadr
                dof_cvelstart[d] = dof_parentid[j.dofadr] if t < 3 else j.dofadr + 2
            elif j.type == JNT_BALL:
                dof_cvelstart[d] = dof_parentid[j.dofadr]
            else:
                dof_cvelstart[d] = dof_parentid[d]
        chainmask = np.zeros((nbk, 2), np.int64)
        for k in range(1, nbk):
            d = int(body_lastdof[k])
            msk = 0
            while d >= 0:
                msk |= 1 << d
                d = int(dof_parentid[d])
            chainmask[k] = [msk & 0xFFFFFFFF, (msk >> 32) & 0xFFFFFFFF]
        chainmask = chainmask.astype(np.uint32).view(np.int32) if False else np.array(
            [[(v if v < 2 ** 31 else v - 2 ** 32) for v in row] for row in chainmask], np.int32)
        body_ancadr = np.zeros(nbk, np.int32)
        body_ancnum = np.zeros(nbk, np.int32)
        body_anc = []
        for k in range(nbk):
            body_ancadr[k] = len(body_anc)
            a_ = int(body_parent[k]) if k > 0 else 0
            while a_ > 0:
                body_anc.append(a_)
                a_ = int(body_parent[a_])
            body_ancnum[k] = len(body_anc) - body_ancadr[k]
        if nbk > 64:
            raise ValueError("engine limit: at most 64 bodies after static-body fusing (subtree masks are 64-bit)")
        submask = np.zeros((nbk, 2), np.int64)
        for k in range(1, nbk):
            msk = 0
            for e in subs[k]:
                msk |= 1 << e
            submask[k] = [msk & 0xFFFFFFFF, (msk >> 32) & 0xFFFFFFFF]
        submask = np.array([[(v if v < 2 ** 31 else v - 2 ** 32) for v in row] for row in submask], np.int32)
        # pointer jumping: a body whose pose needs its own ancestors (not mocap, not a free-joint root) reaches the ancestor
        # 2^s levels up in round s; bodies with world poses already known (mocap / free) never jump
        def _needs_chain(k):
            if k == 0 or body_mocapid[k] >= 0:
                return False
            return not (body_jntnum[k] == 1 and jnts[body_jntadr[k]].type == JNT_FREE)
        njump = max(1, int(np.ceil(np.log2(max(maxdepth, 1) + 1))))
        jump = np.zeros((njump, nbk), np.int32)
        for k in range(1, nbk):
            if _needs_chain(k):
                jump[0, k] = body_parent[k]
        for s_ in range(1, njump):
            for k in range(1, nbk):
                a_ = jump[s_ - 1, k]
                jump[s_, k] = jump[s_ - 1, a_] if a_ > 0 else 0
        # welds: static dof span (union of the two body chains) and Jacobian-pool offset of every active weld
        weld_eq, weld_row, woff = [], [], 0
        for q, e in enumerate(eqs):
            if e["active"] and e["type"] == 1:
                msk = 0
                for bb in (e["obj1"], e["obj2"]):
                    msk |= (int(chainmask[bb][0]) & 0xFFFFFFFF) | ((int(chainmask[bb][1]) & 0xFFFFFFFF) << 32)
                lo = (msk & -msk).bit_length() - 1 if msk else 0
                ln = msk.bit_length() - lo if msk else 0
                weld_eq.append(q)
                weld_row.append(woff | (lo << 12) | (ln << 20))
                woff += 6 * ln
        # joint equalities: one row each over the dof span [min(dof1, dof2), max]; their rows follow the welds' in the row table and in the pool
        jeq_eq, jeq_row, jeq_dof, jeq_qadr = [], [], [], []
        for q, e in enumerate(eqs):
            if e["active"] and e["type"] == EQ_JOINT:
                d1, d2 = e["dofs"]
                lo, ln = min(d1, d2), abs(d1 - d2) + 1
                jeq_eq.append(q); jeq_row.append(woff | (lo << 12) | (ln << 20)); jeq_dof += [d1, d2]; jeq_qadr += list(e["qadrs"])
                woff += ln
        if jeq_eq and any(e["active"] and e["type"] == EQ_WELD and q > jeq_eq[0] for q, e in enumerate(eqs)):
            raise NotImplementedError("welds listed after joint equalities (the row table keeps welds first)")
        T.update(
            jeq_eq=np.array(jeq_eq, np.int32), jeq_row=np.array(jeq_row, np.int32), jeq_dof=np.array(jeq_dof, np.int32), jeq_qadr=np.array(jeq_qadr, np.int32),
            weld_eq=np.array(weld_eq, np.int32), weld_row=np.array(weld_row, np.int32),
            body_submask=submask, body_jump=jump.reshape(-1),
            body_ancadr=body_ancadr, body_ancnum=body_ancnum, body_anc=np.array(body_anc, np.int32),
            body_order=np.array(order, np.int32), level_adr=level_adr, body_subadr=body_subadr, body_subnum=body_subnum,
            body_sub=np.array(body_sub, np.int32), body_lastdof=body_lastdof, mpair_i=np.array(mpi, np.int32),
            mpair_j=np.array(mpj, np.int32), dof_cvelstart=dof_cvelstart, dof_chainmask=chainmask,
            devpair=np.nonzero(pair_supported)[0].astype(np.int32),
        )
        # dof spans of every pair's Jacobian rows: the union of the two body chains, split at its widest run of unused dofs when
        # that run is at least two dofs wide (finger + free object of the hand: wrist..finger | object instead of all 30 dofs)
        def _spans(msk):
            if not msk:
                return 0
            lo, hi = (msk & -msk).bit_length() - 1, msk.bit_length() - 1
            best, run0, d = (0, 0), None, lo
            while d <= hi:
                if not (msk >> d) & 1:
                    run0 = d if run0 is None else run0
                else:
                    if run0 is not None and d - run0 > best[0]:
                        best = (d - run0, run0)
                    run0 = None
                d += 1
            if best[0] >= 2 and c.capacity.get("split_spans", True):
                a_hi, b_lo = best[1] - 1, best[1] + best[0]
                return lo | ((a_hi - lo + 1) << 8) | (b_lo << 16) | ((hi - b_lo + 1) << 24)
            return lo | ((hi - lo + 1) << 8)

        def _mask_of(bid):
            return (int(chainmask[bid][0]) & 0xFFFFFFFF) | ((int(chainmask[bid][1]) & 0xFFFFFFFF) << 32)

        pair_span = np.zeros(npair, np.int64)
        for pi_ in range(npair):
            pair_span[pi_] = _spans(_mask_of(int(geom_bodyid[pair_geom1[pi_]])) | _mask_of(int(geom_bodyid[pair_geom2[pi_]])))
        T["pair_span"] = pair_span.astype(np.uint32).view(np.int32) if npair else np.zeros(0, np.int32)
        if ng >= 4096:
            raise ValueError("engine limit: at most 4095 geoms (packed candidate records)")
        # Wall lattice (maze layouts): world-fixed, axis-aligned boxes of one size on a regular xy grid.  A moving sphere / capsule can only
        # touch the walls of the 3 x 3 cells around it, so its wall pairs leave the flat candidate list (819 sphere tests per pass for
        # AntMaze_Large) and are looked up through the cell table instead: grid_param = (x0, y0, 1 / cell, nx, ny), grid_cell[iy * nx + ix] =
        # wall number or -1, grid_wall_geom[wall] = geom id, grid_geom[k] = the k-th moving geom (| type << 12), grid_geom_bound[k] = (broad-phase radius incl. margin, margin), grid_pair[k * nwall + wall] = pair index.
        dp = np.asarray(T["devpair"]).astype(np.int64)
        T["grid_param"], T["grid_cell"], T["grid_wall_geom"], T["grid_geom"], T["grid_pair"] = np.zeros(0), np.zeros(0, np.int32), np.zeros(0, np.int32), np.zeros(0, np.int32), np.zeros(0, np.int32)
        T["grid_geom_bound"] = np.zeros(0)
        ident = np.array([1.0, 0.0, 0.0, 0.0])
        walls = [gi for gi in range(ng) if geom_type[gi] == GEOM_BOX and geom_bodyid[gi] == 0 and np.abs(np.abs(geom_quat[gi]) - ident).max() < 1e-12]
        if len(walls) >= 8:
            sz = geom_size[walls]
            cell = 2.0 * sz[0, 0]
            same = np.abs(sz - sz[0]).max() < 1e-12 and abs(sz[0, 0] - sz[0, 1]) < 1e-12 and np.abs(geom_pos[walls][:, 2] - geom_pos[walls[0]][2]).max() < 1e-12
            xy = geom_pos[walls][:, :2]
            ij = (xy - xy.min(axis=0)) / cell
            if same and np.abs(ij - np.round(ij)).max() < 1e-9:
                ij = np.round(ij).astype(int)
                nx_, ny_ = int(ij[:, 0].max()) + 1, int(ij[:, 1].max()) + 1
                cellmap = -np.ones((ny_, nx_), np.int32)
                for w_, (ix_, iy_) in enumerate(ij):
                    cellmap[iy_, ix_] = w_
                wall_no = {gi: w_ for w_, gi in enumerate(walls)}
                pair_of = {}
                for pi_ in dp:
                    a_, b_ = int(pair_geom1[pi_]), int(pair_geom2[pi_])
                    if b_ in wall_no and geom_type[a_] in (GEOM_SPHERE, GEOM_CAPSULE) and geom_bodyid[a_] != 0:
                        pair_of[(a_, wall_no[b_])] = int(pi_)
                movers = sorted({a_ for a_, _ in pair_of})
                movers = [a_ for a_ in movers if all((a_, w_) in pair_of for w_ in range(len(walls)))]
                # a mover must fit into the 3 x 3 neighbourhood: bounding radius + margin below half a cell
                movers = [a_ for a_ in movers if geom_rbound[a_] + max(pair_margin[pair_of[(a_, w_)]] for w_ in range(len(walls))) < 0.5 * cell]
                if movers:
                    x0_, y0_ = xy.min(axis=0) - 0.5 * cell
                    T["grid_param"] = np.array([x0_, y0_, 1.0 / cell, nx_, ny_], dtype=np.float64)
                    T["grid_cell"], T["grid_wall_geom"] = cellmap.reshape(-1), np.array(walls, np.int32)
                    T["grid_geom"] = np.array([a_ | (int(geom_type[a_]) << 12) for a_ in movers], np.int32)      # geom id | type << 12
                    # per mover: broad-phase radius (its own + the walls' + margin) and the pair margin (one value per mover: checked)
                    mg = [sorted({float(pair_margin[pair_of[(a_, w_)]]) for w_ in range(len(walls))}) for a_ in movers]
                    assert all(len(x_) == 1 for x_ in mg)
                    T["grid_geom_bound"] = np.array([[geom_rbound[a_] + geom_rbound[walls[0]] + mg[k_][0], mg[k_][0]] for k_, a_ in enumerate(movers)], dtype=np.float64).reshape(-1)
                    T["grid_pair"] = np.array([[pair_of[(a_, w_)] for w_ in range(len(walls))] for a_ in movers], np.int32).reshape(-1)
                    gridded = set(T["grid_pair"].tolist())
                    dp = np.array([pi_ for pi_ in dp if int(pi_) not in gridded], dtype=np.int64)
                    T["devpair"] = dp.astype(np.int32)
                    info["grid"] = dict(nx=nx_, ny=ny_, walls=len(walls), movers=len(movers), pairs_left=len(dp))
        g1s, g2s = pair_geom1[dp], pair_geom2[dp]
        T["devpair_geoms"] = (g1s | (g2s << 12) | (geom_type[g1s] << 24) | (geom_type[g2s] << 28)).astype(np.int64).astype(np.uint32).view(np.int32) if len(dp) else np.zeros(0, np.int32)
        T["devpair_bound"] = np.stack([pair_margin[dp], np.where(geom_type[g1s] == GEOM_PLANE, geom_rbound[g2s], geom_rbound[g1s] + geom_rbound[g2s])], axis=1) if len(dp) else np.zeros((0, 2))
        # joint-box gates of the hull pairs (mjcf/pair_gates.py; requested per model family: the analysis takes a minute or two)
        T["devpair_gate"], T["gate_qadr"], T["gate_box"] = -np.ones(len(dp), np.int32), np.zeros(0, np.int32), np.zeros(0)
        if c.capacity.get("pair_gates") and len(dp):
            from .pair_gates import compute_pair_gates

            T["devpair_gate"], T["gate_qadr"], T["gate_box"], rep = compute_pair_gates(T)
            info["pair_gates"] = [dict(pair=p_, geoms=[g1_, g2_], joints=j_, box=b_, slack=s_) for p_, g1_, g2_, j_, b_, s_ in rep]
        info["nmpair"] = len(mpi)
        info["nbody_full"] = nb
        info["unsupported_pairs"] = int(np.sum(pair_supported == 0))
        info["origin"] = c.origin.tolist()
        return CompiledModel(T, names, info)


def compile_mjcf(xml_path: str, mutate=None, capacity=None, touch_filter=None, keep_sites=None, shift_body=None, shift_rotates=False, origin=None) -> CompiledModel:
    """capacity: optional {"maxefc": rows, "jpool": words, "maxcon": contacts, "split_spans": bool} request for the engine's per-world constraint tables.
    touch_filter: optional callable(sensor name) selecting the <touch> sensors the engine evaluates.
    keep_sites: optional list of site names whose world frames the engine tracks (default: every site of the model)."""
    return MjcfCompiler(xml_path, mutate=mutate, capacity=capacity, touch_filter=touch_filter, keep_sites=keep_sites, shift_body=shift_body, shift_rotates=shift_rotates, origin=origin).compile()
