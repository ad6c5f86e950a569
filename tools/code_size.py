#!/usr/bin/env python3
"""Code bytes of every gfx950 function in the objects of acvm_amd/build (kernels and out-of-line device functions), largest first: the instruction
cache is 64 KiB per two CUs, and a kernel whose hot loop does not fit runs from L2 (NOTEBOOK.md section 9, round 4).  python tools/code_size.py [min KiB]"""
import glob
import os
import re
import subprocess
import sys
import tempfile

LLVM = "/opt/rocm/lib/llvm/bin"
ROOT = os.environ.get("ACVM_ROOT") or os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
floor = int(sys.argv[1]) if len(sys.argv) > 1 else 16
rows = {}
for obj in sorted(glob.glob(os.path.join(ROOT, "acvm_amd", "build", "*.hip.o"))):
    with tempfile.TemporaryDirectory() as d:
        subprocess.run([f"{LLVM}/llvm-objcopy", "-O", "binary", "--only-section=.hip_fatbin", obj, f"{d}/fat.bin"], check=True)
        raw = open(f"{d}/fat.bin", "rb").read()
        pos = [m.start() for m in re.finditer(b"__CLANG_OFFLOAD_BUNDLE__", raw)]
        for a, b in zip(pos, pos[1:] + [len(raw)]):
            open(f"{d}/fat1", "wb").write(raw[a:b])
            r = subprocess.run([f"{LLVM}/clang-offload-bundler", "--unbundle", "--type=o", f"--input={d}/fat1", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950",
                                f"--output={d}/co"], capture_output=True)
            if r.returncode or not os.path.exists(f"{d}/co"):
                continue
            out = subprocess.run([f"{LLVM}/llvm-readelf", "-s", "--wide", f"{d}/co"], capture_output=True, text=True).stdout
            for line in out.splitlines():
                p = line.split()
                if len(p) >= 8 and p[3] == "FUNC" and int(p[2]) >= floor * 1024:
                    name = subprocess.run(["c++filt", p[7]], capture_output=True, text=True).stdout.strip()
                    rows[(os.path.basename(obj), re.sub(r"\(.*", "", name))] = int(p[2])
            os.remove(f"{d}/co")
for (obj, name), size in sorted(rows.items(), key=lambda kv: -kv[1]):
    print(f"{size / 1024:8.1f} KiB  {obj:24s} {name[:120]}")
