"""Loader for the HIP extension (libgrx_hip.so, C ABI in include/grx_capi.h).

There is deliberately NO fallback: if the extension is missing or no MI355X is visible the
product path raises.  (The fp64 oracle under oracle/ is test infrastructure and is never
imported from here.)
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("GRX_HIP_LIB") or os.path.join(_HERE, "_lib", "libgrx_hip.so")   # $GRX_HIP_LIB: an alternative BUILD of the same library (A/B of compiler options); never a fallback
_lib = None


class OverflowLaneStruct(ctypes.Structure):
    """mirrors grx_overflow_lane (include/grx_capi.h)"""
    _fields_ = [(n, ctypes.c_void_p) for n in ("skip", "entry_count", "entry_list", "list", "count", "next_flags", "next_count", "next_list", "ttl")] + [
        (n, ctypes.c_int) for n in ("soft_maxefc", "soft_jpool", "soft_maxcon", "ttl_init", "grid", "entry_cap", "next_cap")] + [
        (n, ctypes.c_void_p) for n in ("ready", "progress", "poll_list")] + [(n, ctypes.c_int) for n in ("ready_cap", "progress_total", "poll_grid", "pad_")]


class FetchBuffersStruct(ctypes.Structure):
    _fields_ = [(n, ctypes.c_void_p) for n in (
        "qpos", "qvel", "qacc_ws", "mocap", "aux", "goal", "action", "obs", "achieved", "reward", "success", "status", "mask", "order", "cost", "packed", "hullcache", "handoff")] + [
        ("handoff_stride", ctypes.c_int), ("handoff_large", ctypes.c_int), ("split_state", ctypes.c_void_p), ("split_parts", ctypes.c_int), ("split_pad_", ctypes.c_int), ("lane", OverflowLaneStruct)]


class FetchResetArgsStruct(ctypes.Structure):
    _fields_ = [(n, ctypes.c_void_p) for n in ("idx", "samples", "init_qpos", "init_qvel", "init_mocap")] + [("obj_qadr", ctypes.c_int), ("keep_outcome", ctypes.c_int), ("final_packed", ctypes.c_void_p)]


class PointTaskStruct(ctypes.Structure):
    _fields_ = [("n_substeps", ctypes.c_int), ("sparse_reward", ctypes.c_int), ("continuing_task", ctypes.c_int), ("agent", ctypes.c_int),
                ("goal_radius", ctypes.c_double), ("vel_clip", ctypes.c_float)]


class PointBuffersStruct(ctypes.Structure):
    _fields_ = [(n, ctypes.c_void_p) for n in (
        "qpos", "qvel", "qacc_ws", "goal", "action", "obs", "achieved", "reward", "success", "terminated", "status", "mask", "packed", "split_state")] + [
        ("split_parts", ctypes.c_int), ("split_pad_", ctypes.c_int)]


class AdroitTaskStruct(ctypes.Structure):
    _fields_ = [("n_substeps", ctypes.c_int), ("sparse_reward", ctypes.c_int), ("kind", ctypes.c_int), ("site", ctypes.c_int * 5), ("obj_body", ctypes.c_int),
                ("nq_obs", ctypes.c_int), ("obs_dim", ctypes.c_int), ("qadr", ctypes.c_int * 2), ("len", ctypes.c_float * 2)]


class AdroitBuffersStruct(ctypes.Structure):
    _fields_ = [(n, ctypes.c_void_p) for n in ("qpos", "qvel", "qacc_ws", "shift", "target", "action", "act_mean", "act_rng", "obs", "reward", "success", "status", "mask")] + [("lane", OverflowLaneStruct), ("compact", ctypes.c_void_p), ("n_compact", ctypes.c_int), ("order", ctypes.c_void_p), ("cost", ctypes.c_void_p), ("split_rows", ctypes.c_void_p), ("split_state", ctypes.c_void_p), ("split_stride", ctypes.c_int), ("split_parts", ctypes.c_int)]


class KitchenTaskStruct(ctypes.Structure):
    _fields_ = [("n_substeps", ctypes.c_int), ("obs_dim", ctypes.c_int), ("dt", ctypes.c_float), ("vel_lo", ctypes.c_float * 9), ("vel_hi", ctypes.c_float * 9),
                ("pos_lo", ctypes.c_float * 9), ("pos_hi", ctypes.c_float * 9), ("noise_scale", ctypes.c_float * 59), ("task_adr", ctypes.c_int * 7),
                ("task_num", ctypes.c_int * 7), ("task_goal", ctypes.c_float * 17), ("bonus_thresh", ctypes.c_float)]


class KitchenBuffersStruct(ctypes.Structure):
    _fields_ = [(n, ctypes.c_void_p) for n in ("qpos", "qvel", "qacc_ws", "last_qpos", "action", "noise", "obs", "completed", "status", "mask", "skin")] + [
        ("skin_stride", ctypes.c_int), ("skin_radius", ctypes.c_float), ("order", ctypes.c_void_p), ("cost", ctypes.c_void_p), ("lane", OverflowLaneStruct),
        ("split_rows", ctypes.c_void_p), ("split_state", ctypes.c_void_p), ("split_stride", ctypes.c_int), ("split_parts", ctypes.c_int)]


class KitchenBookStruct(ctypes.Structure):      # include/grx_capi.h, grx_kitchen_book
    _fields_ = [(n, ctypes.c_void_p) for n in ("completed", "stepped", "tasks_to_complete", "episode_completions", "elapsed", "step_completions", "reward", "terminated",
                                               "truncated", "needs_reset", "reset_now", "qpos", "qvel", "qacc_ws", "init_qpos")] + [
        (n, ctypes.c_int) for n in ("nq", "nv", "all_mask", "max_steps", "remove_when_completed", "terminate_when_completed", "mode")] + [("final_info", ctypes.c_void_p)]


class HerArgsStruct(ctypes.Structure):
    _fields_ = [("rows", ctypes.c_void_p), ("acts", ctypes.c_void_p)] + [(n, ctypes.c_int) for n in ("T", "N", "W", "obs_dim", "goal_dim", "act_dim")] + [
        (n, ctypes.c_void_p) for n in ("t_idx", "w_idx", "t_goal")] + [("kind", ctypes.c_int), ("p0", ctypes.c_double), ("p1", ctypes.c_double)] + [
        (n, ctypes.c_int) for n in ("sparse", "ignore_pos", "ignore_rot", "ignore_z")] + [("out", ctypes.c_void_p), ("term_rows", ctypes.c_void_p), ("term_t", ctypes.c_void_p)]


class MazeResetArgsStruct(ctypes.Structure):
    _fields_ = [(n, ctypes.c_void_p) for n in ("idx", "stage", "qpos0")] + [(n, ctypes.c_int) for n in ("nq", "nv", "obs_dim", "obs_skip")] + [
        ("goal_radius", ctypes.c_double), ("keep_outcome", ctypes.c_int)] + [
        (n, ctypes.c_void_p) for n in ("qpos", "qvel", "qacc_ws", "goal", "obs", "achieved", "reward", "success", "packed")]


class HandCommitArgsStruct(ctypes.Structure):
    _fields_ = [("idx", ctypes.c_void_p), ("k", ctypes.c_int)] + [(n, ctypes.c_int) for n in ("nq", "nv", "obs_dim", "goal_dim")] + [
        (n, ctypes.c_void_p) for n in ("s_qpos", "s_qvel", "s_qacc_ws", "s_obs", "s_achieved", "s_palm", "s_goal", "s_packed", "s_status",
                                       "qpos", "qvel", "qacc_ws", "obs", "achieved", "palm", "goal", "packed", "status")]


class FetchCommitArgsStruct(ctypes.Structure):      # include/grx_capi.h, grx_fetch_commit_args
    _fields_ = [("idx", ctypes.c_void_p), ("k", ctypes.c_int)] + [(n, ctypes.c_int) for n in ("nq", "nv", "mocap_words", "obs_dim")] + [
        (n, ctypes.c_void_p) for n in ("s_qpos", "s_qvel", "s_qacc_ws", "s_mocap", "s_aux", "s_goal", "s_obs", "s_achieved", "s_status",
                                       "qpos", "qvel", "qacc_ws", "mocap", "aux", "goal", "obs", "achieved", "packed", "final_packed", "status")]


class AdroitCommitArgsStruct(ctypes.Structure):      # include/grx_capi.h, grx_adroit_commit_args
    _fields_ = [("idx", ctypes.c_void_p), ("k", ctypes.c_int)] + [(n, ctypes.c_int) for n in ("nq", "nv", "obs_dim")] + [
        (n, ctypes.c_void_p) for n in ("s_qpos", "s_qvel", "s_qacc_ws", "s_shift", "s_target", "s_obs", "s_status", "qpos", "qvel", "qacc_ws", "shift", "target", "obs", "status")]


class HandBuffersStruct(ctypes.Structure):
    _fields_ = [(n, ctypes.c_void_p) for n in (
        "qpos", "qvel", "qacc_ws", "goal", "action", "obs", "achieved", "palm", "reward", "success", "status", "mask", "order", "cost", "packed")] + [("lane", OverflowLaneStruct),
        ("split_rows", ctypes.c_void_p), ("split_state", ctypes.c_void_p), ("split_stride", ctypes.c_int), ("split_parts", ctypes.c_int)]


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"HIP extension not built: {LIB_PATH} is missing. Run `python -c 'import __graft_entry__ as g; g.build()'` "
                "(hipcc --offload-arch=gfx950). There is no CPU fallback.")
        import torch  # noqa: F401  -- first: the library must bind to the HIP runtime torch has already loaded (loaded the other way round, a second runtime instance sees no device)

        L = ctypes.CDLL(LIB_PATH)
        vp, ci = ctypes.c_void_p, ctypes.c_int
        L.grx_last_error.restype = ctypes.c_char_p
        L.grx_model_create.argtypes = [vp, ci, vp, ci, vp, ci, ci, ctypes.POINTER(vp)]
        L.grx_model_destroy.argtypes = [vp]
        L.grx_model_set_table.argtypes = [vp, ctypes.c_char_p, vp, ci]
        L.grx_model_lds_bytes.argtypes = [vp]
        L.grx_model_dim.argtypes = [vp, ctypes.c_char_p]
        L.grx_fetch_step.argtypes = [vp, vp, vp, ci, vp]
        L.grx_fetch_forward.argtypes = [vp, vp, vp, ci, ci, vp]
        L.grx_fetch_reset.argtypes = [vp, vp, vp, vp, ci, vp]
        L.grx_fetch_compute_reward.argtypes = [vp, vp, ctypes.c_int64, ctypes.c_double, ci, vp, vp]
        L.grx_her_relabel.argtypes = [vp, ctypes.c_int64, vp]
        L.grx_her_sample.argtypes = [vp, ci, ci, ci, ci, ctypes.c_uint64, ctypes.c_uint64, ctypes.c_int64, vp, vp, vp, vp]
        L.grx_her_sample_final.argtypes = [vp, vp, vp, ci, ci, ci, ci, ctypes.c_uint64, ctypes.c_uint64, ctypes.c_int64, vp, vp, vp, vp]
        L.grx_her_mark_resets.argtypes = [vp, ci, ci, vp, vp, vp, vp]
        L.grx_kitchen_step.argtypes = [vp, vp, vp, ci, ci, vp]
        L.grx_sample_uniform_rows.argtypes = [vp, vp, ci, ci, vp]
        L.grx_uniform_rows_device.argtypes = [vp, vp, ci, ci, vp, vp]
        L.grx_kitchen_bookkeeping.argtypes = [vp, ci, vp]
        L.grx_point_step.argtypes = [vp, vp, vp, ci, vp]
        L.grx_maze_compute_reward.argtypes = [vp, vp, ctypes.c_int64, ctypes.c_double, ci, vp, vp]
        L.grx_hand_step.argtypes = [vp, vp, vp, ci, ci, vp]
        L.grx_hand_step_repeat.argtypes = [vp, vp, vp, ci, ci, vp]
        L.grx_adroit_step.argtypes = [vp, vp, vp, ci, ci, vp]
        L.grx_goal_compute_reward.argtypes = [vp, vp, ctypes.c_int64, ci, ctypes.c_double, ci, vp, vp]
        L.grx_manip_compute_reward.argtypes = [vp, vp, ctypes.c_int64, ci, ci, ci, ctypes.c_float, ctypes.c_float, ci, vp, vp]
        L.grx_order_by_cost.argtypes = [vp, vp, ctypes.c_float, ci, vp, vp]
        L.grx_order_by_cost_slots.argtypes = [vp, vp, ctypes.c_float, ci, ci, vp, vp]
        L.grx_maze_reset_rows.argtypes = [vp, ci, vp]
        L.grx_hand_commit_rows.argtypes = [vp, vp]
        L.grx_fetch_commit_rows.argtypes = [vp, vp]
        L.grx_adroit_commit_rows.argtypes = [vp, vp]
        cd = ctypes.c_double
        L.grx_fetch_sample_resets.argtypes = [vp, vp, ci, ci, ci, cd, cd, vp, vp, cd, vp, vp]
        L.grx_fetch_sample_resets_device.argtypes = [vp, vp, ci, ci, ci, cd, cd, vp, vp, cd, vp, vp]
        L.grx_adroit_sample_resets_device.argtypes = [vp, vp, ci, ci, vp, vp, vp, vp, vp, vp]
        L.grx_maze_sample_resets_device.argtypes = [vp, vp, ci, vp, ci, vp, ci, cd, cd, vp, vp, vp, vp]
        _lib = L
    return _lib


def build_id(path: str = None) -> str:
    """Identity of the DEVICE code inside a build of libgrx_hip.so: sha256 (first 16 hex digits) of its `.hip_fatbin` section -- the bundle of gfx950 code objects -- read with a
    minimal ELF64 section walk (no tools, no GPU).  The profile summaries bench.py quotes (profiles/pmc_*.json, tools/collect_profiles.py) carry the id of the build they were
    measured on; bench.py attaches them only while it matches the library that is actually loaded."""
    import hashlib
    import struct

    with open(path or LIB_PATH, "rb") as f:
        blob = f.read()
    if blob[:4] != b"\x7fELF" or blob[4] != 2:
        raise RuntimeError("libgrx_hip.so is not an ELF64 file")
    shoff, = struct.unpack_from("<Q", blob, 0x28)
    shentsize, shnum, shstrndx = struct.unpack_from("<HHH", blob, 0x3A)
    sec = lambda i: struct.unpack_from("<IIQQQQIIQQ", blob, shoff + i * shentsize)      # name, type, flags, addr, offset, size, link, info, addralign, entsize
    stroff = sec(shstrndx)[4]
    for i in range(shnum):
        name, _, _, _, off, size = sec(i)[:6]
        end = blob.index(b"\0", stroff + name)
        if blob[stroff + name: end] == b".hip_fatbin":
            return hashlib.sha256(blob[off: off + size]).hexdigest()[:16]
    raise RuntimeError("libgrx_hip.so holds no .hip_fatbin section (not a HIP build)")


def check(rc: int):
    if rc != 0:
        raise RuntimeError("libgrx_hip: " + lib().grx_last_error().decode())


EXPORTED_SYMBOLS = [
    "grx_model_create", "grx_model_destroy", "grx_model_set_table", "grx_model_lds_bytes", "grx_model_dim",
    "grx_fetch_step", "grx_fetch_forward", "grx_fetch_reset", "grx_fetch_compute_reward", "grx_her_relabel", "grx_her_sample", "grx_her_sample_final", "grx_her_mark_resets", "grx_fetch_sample_resets", "grx_fetch_sample_resets_device", "grx_adroit_sample_resets_device", "grx_maze_sample_resets_device", "grx_point_step", "grx_maze_compute_reward", "grx_hand_step", "grx_hand_step_repeat", "grx_adroit_step", "grx_kitchen_step", "grx_sample_uniform_rows", "grx_uniform_rows_device", "grx_kitchen_bookkeeping", "grx_goal_compute_reward", "grx_manip_compute_reward", "grx_order_by_cost", "grx_order_by_cost_slots", "grx_maze_reset_rows", "grx_hand_commit_rows", "grx_fetch_commit_rows", "grx_adroit_commit_rows", "grx_last_error",
]


# ---------------------------------------------------------------------------------------------------- shared model handles
# The native side has GRX_MAX_MODELS (32) descriptor slots per process (constant memory, csrc/grx_engine.h).  Environments never modify a model after creating it
# (per-world model edits are per-world state), so environments built from the same compiled tables SHARE one handle: an id costs two slots (fast + overflow-lane
# tables) however many environments of it a process holds, and the 17th environment no longer fails with "all model descriptor slots are in use".
_MODEL_LOCK = __import__("threading").Lock()
_MODELS = {}      # (device index, digest of the packed tables) -> [handle, reference count]
_MODEL_KEYS = {}  # handle value -> key


def acquire_model(H, I, F, device_index):
    """handle of the model packed as (H, I, F) on `device_index`: created on first use, reference-counted afterwards (release_model)"""
    import hashlib

    # (GRX_NO_HULLCELLS changes what grx_model_create derives from the tables -- the A/B switch of the hulls' support-candidate lists -- so it is part of the key)
    key = (int(device_index), hashlib.sha1(H.tobytes() + I.tobytes() + F.tobytes()).hexdigest(), os.environ.get("GRX_NO_HULLCELLS") is not None)
    with _MODEL_LOCK:
        ent = _MODELS.get(key)
        if ent is None:
            h = ctypes.c_void_p()
            check(lib().grx_model_create(H.ctypes.data, H.size, I.ctypes.data, I.size, F.ctypes.data, F.size, int(device_index), ctypes.byref(h)))
            ent = _MODELS[key] = [h, 0]
            _MODEL_KEYS[h.value] = key
        ent[1] += 1
        return ent[0]


def release_model(h):
    """drop one reference; the native model is destroyed with the last one"""
    if h is None or not getattr(h, "value", None):
        return
    with _MODEL_LOCK:
        key = _MODEL_KEYS.get(h.value)
        if key is None:
            return
        ent = _MODELS[key]
        ent[1] -= 1
        if ent[1] <= 0:
            del _MODELS[key], _MODEL_KEYS[h.value]
            lib().grx_model_destroy(h)
