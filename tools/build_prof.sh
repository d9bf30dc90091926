#!/bin/sh
# rebuild the -DGRX_PROFILE variant of the HIP library used by tools/profile_stages.py
cd "$(dirname "$0")/.." && python - <<'PY'
import subprocess
from __graft_entry__ import HIPCC_FLAGS
subprocess.check_call(["/opt/rocm/bin/hipcc"] + HIPCC_FLAGS + ["-DGRX_PROFILE", "-o", "gymnasium_robotics_amd/_lib/libgrx_hip_prof.so", "gymnasium_robotics_amd/csrc/grx_kernels.hip"])
PY
