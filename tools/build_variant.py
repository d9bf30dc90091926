"""A/B builds of the HIP library that differ in a few translation units only: the named units are compiled with the extra flags, every other object comes from the
main build (gymnasium_robotics_amd/_lib/.obj_libgrx_hip), so a Fetch-only experiment costs one compile instead of six.

    python tools/build_variant.py <name> "<flags>" FETCH [HAND ...]      -> gymnasium_robotics_amd/_lib/libgrx_hip_<name>.so
"""
import os
import subprocess
import sys

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)
from __graft_entry__ import HIP_SRC, HIP_UNITS, HIPCC_FLAGS  # noqa: E402

name, flags, units = sys.argv[1], sys.argv[2].split(), [u.upper() for u in sys.argv[3:]]
lib = os.path.join(ROOT, "gymnasium_robotics_amd", "_lib")
main_obj, var_obj = os.path.join(lib, ".obj_libgrx_hip"), os.path.join(lib, f".obj_libgrx_hip_{name}")
os.makedirs(var_obj, exist_ok=True)
cflags = [f for f in HIPCC_FLAGS if f != "-shared"] + flags
procs = []
for u in units:
    obj = os.path.join(var_obj, f"grx_{u.lower()}.o")
    procs.append((u, obj, subprocess.Popen(["/opt/rocm/bin/hipcc"] + cflags + [f"-DGRX_TU_{u}=1", "-c", "-o", obj, HIP_SRC], stderr=subprocess.DEVNULL)))
objs = {u: os.path.join(main_obj, f"grx_{u.lower()}.o") for u in HIP_UNITS}
for u, obj, p in procs:
    assert p.wait() == 0, u
    objs[u] = obj
out = os.path.join(lib, f"libgrx_hip_{name}.so")
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", out] + list(objs.values()))
print(out)
