"""Register / scratch / LDS budget of every kernel in libgrx_hip.so, read from the code objects' metadata notes
(llvm-readelf --notes): what DESIGN.md's occupancy statements have to agree with.

    python tools/kernel_resources.py [path/to/lib.so]
"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
LLVM = "/opt/rocm/lib/llvm/bin"


def kernel_resources(so):
    """so: the library or any of its objects (every .hip_fatbin bundle inside is read)"""
    out = []
    with tempfile.TemporaryDirectory() as d:
        fat = os.path.join(d, "fatbin")
        subprocess.run([os.path.join(LLVM, "llvm-objcopy"), "--dump-section", f".hip_fatbin={fat}", so, os.path.join(d, "copy")], capture_output=True)
        blob = open(fat, "rb").read()
        # the section of a linked library holds one bundle per translation unit, each starting with the bundler magic
        starts = [m.start() for m in re.finditer(b"__CLANG_OFFLOAD_BUNDLE__", blob)]
        for k, st in enumerate(starts):
            part = os.path.join(d, f"bundle{k}")
            open(part, "wb").write(blob[st: starts[k + 1] if k + 1 < len(starts) else len(blob)])
            co = os.path.join(d, f"co{k}")
            r = subprocess.run([os.path.join(LLVM, "clang-offload-bundler"), "--unbundle", "--type=o", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950",
                                f"--input={part}", f"--output={co}"], capture_output=True)
            if r.returncode or not os.path.exists(co):
                continue
            notes = subprocess.run([os.path.join(LLVM, "llvm-readelf"), "--notes", co], capture_output=True, text=True).stdout
            for blk in notes.split("- .agpr_count:")[1:]:
                g = lambda key: (re.search(rf"\.{key}:\s*(\S+)", blk) or [None, "?"])[1]
                out.append(dict(name=g("name"), vgpr=g("vgpr_count"), agpr=blk.split()[0], sgpr=g("sgpr_count"), scratch=g("private_segment_fixed_size"),
                                lds=g("group_segment_fixed_size"), spill_v=g("vgpr_spill_count"), spill_s=g("sgpr_spill_count")))
    return out


if __name__ == "__main__":
    so = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "gymnasium_robotics_amd", "_lib", "libgrx_hip.so")
    rows = kernel_resources(so)
    demangle = subprocess.run(["c++filt"] + [r["name"] for r in rows], capture_output=True, text=True).stdout.splitlines()
    print(f"{'kernel':110s} vgpr agpr sgpr scratch vspill sspill")
    for r, n in sorted(zip(rows, demangle), key=lambda x: x[1]):
        n = re.sub(r"GrxShape<([^>]*)>", lambda m: "S<" + m.group(1).replace(" ", "") + ">", n)
        print(f"{n[:110]:110s} {r['vgpr']:>4} {r['agpr']:>4} {r['sgpr']:>4} {r['scratch']:>7} {r['spill_v']:>6} {r['spill_s']:>6}")
