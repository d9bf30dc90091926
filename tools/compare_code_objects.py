"""Are the device code objects of two builds of libgrx_hip.so the same machine code?  (round 5: the engine header was split into stage fragments -- a purely textual change.)
Every gfx950 code object inside the two libraries' .hip_fatbin sections is unbundled and its .text section compared byte for byte, translation unit by translation unit.

    python tools/compare_code_objects.py old.so new.so
"""
import hashlib
import os
import re
import subprocess
import sys
import tempfile

LLVM = "/opt/rocm/lib/llvm/bin"


def texts(so):
    out = []
    with tempfile.TemporaryDirectory() as d:
        fat = os.path.join(d, "fatbin")
        subprocess.run([os.path.join(LLVM, "llvm-objcopy"), "--dump-section", f".hip_fatbin={fat}", so, os.path.join(d, "copy")], capture_output=True)
        blob = open(fat, "rb").read()
        starts = [m.start() for m in re.finditer(b"__CLANG_OFFLOAD_BUNDLE__", blob)]
        for k, st in enumerate(starts):
            part, co, txt = os.path.join(d, f"b{k}"), os.path.join(d, f"co{k}"), os.path.join(d, f"t{k}")
            open(part, "wb").write(blob[st: starts[k + 1] if k + 1 < len(starts) else len(blob)])
            r = subprocess.run([os.path.join(LLVM, "clang-offload-bundler"), "--unbundle", "--type=o", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", f"--input={part}", f"--output={co}"], capture_output=True)
            if r.returncode or not os.path.exists(co):
                continue
            subprocess.run([os.path.join(LLVM, "llvm-objcopy"), "-O", "binary", "--only-section=.text", co, txt], capture_output=True)
            syms = subprocess.run([os.path.join(LLVM, "llvm-readelf"), "--symbols", "--wide", co], capture_output=True, text=True).stdout
            kern = sorted({ln.split()[-1] for ln in syms.splitlines() if " FUNC " in ln and " GLOBAL " in ln})
            data = open(txt, "rb").read() if os.path.exists(txt) else b""
            out.append((len(data), hashlib.sha256(data).hexdigest()[:16], len(kern), (kern[0][:60] if kern else "")))
    return out


if __name__ == "__main__":
    a, b = texts(sys.argv[1]), texts(sys.argv[2])
    same = 0
    for k, (x, y) in enumerate(zip(a, b)):
        ok = x[:2] == y[:2]
        same += ok
        print(f"code object {k}: {x[2]} kernels, .text {x[0]} B sha {x[1]}  |  {y[2]} kernels, .text {y[0]} B sha {y[1]}  -> {'IDENTICAL' if ok else 'DIFFERENT'}   ({x[3]})")
    print(f"{same} of {max(len(a), len(b))} code objects identical")
