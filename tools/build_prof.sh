#!/bin/sh
# rebuild the -DGRX_PROFILE variant of the HIP library used by tools/profile_stages.py
#   tools/build_prof.sh [fetch|hand|point|adroit|kitchen] [extra -D flags]     (default: fetch)
# The profiled kernel family is compiled together with the C ABI in one unit (the cycle counters live there), the other families
# are compiled without profiling code, all in parallel.
cd "$(dirname "$0")/.." && FAMILY="${1:-fetch}" EXTRA="$2" python - <<'PY'
import os, subprocess
from __graft_entry__ import HIPCC_FLAGS, HIP_SRC
fam = os.environ["FAMILY"].upper()
extra = os.environ.get("EXTRA", "").split()
out = "gymnasium_robotics_amd/_lib/libgrx_hip_prof.so"
objdir = "gymnasium_robotics_amd/_lib/.obj_prof"
os.makedirs(objdir, exist_ok=True)
flags = [f for f in HIPCC_FLAGS if f != "-shared"]
units = [([f"-DGRX_TU_{fam}=1", "-DGRX_TU_API=1", "-DGRX_PROFILE"] + extra, f"{objdir}/prof.o")]
units += [([f"-DGRX_TU_{u}=1"] + extra, f"{objdir}/{u.lower()}.o") for u in ("FETCH", "HAND", "POINT", "ADROIT", "KITCHEN") if u != fam]
procs = [subprocess.Popen(["/opt/rocm/bin/hipcc"] + flags + d + ["-c", "-o", o, HIP_SRC]) for d, o in units]
assert all(p.wait() == 0 for p in procs)
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", out] + [o for _, o in units])
PY
