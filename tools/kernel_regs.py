#!/usr/bin/env python3
"""Register / scratch / LDS usage of every gfx950 kernel in libacvm_amd.so (or the .o files of acvm_amd/build), read from the code
objects' metadata notes: python tools/kernel_regs.py [file ...]"""
import os
import re
import subprocess
import sys
import tempfile

LLVM = "/opt/rocm/lib/llvm/bin"
MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"


def code_objects(path):
    with tempfile.TemporaryDirectory() as d:
        subprocess.run([f"{LLVM}/llvm-objcopy", "-O", "binary", "--only-section=.hip_fatbin", path, f"{d}/fat.bin"], check=True)
        raw = open(f"{d}/fat.bin", "rb").read()
    pos = [m.start() for m in re.finditer(MAGIC, raw)]
    for a, b in zip(pos, pos[1:] + [len(raw)]):
        with tempfile.TemporaryDirectory() as d:
            open(f"{d}/fat", "wb").write(raw[a:b])
            r = subprocess.run([f"{LLVM}/clang-offload-bundler", "--unbundle", "--type=o", f"--input={d}/fat", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950",
                                f"--output={d}/co"], capture_output=True)
            if r.returncode == 0 and os.path.exists(f"{d}/co"):
                yield subprocess.run([f"{LLVM}/llvm-readelf", "--notes", f"{d}/co"], capture_output=True, text=True).stdout


def main():
    files = sys.argv[1:] or [os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "acvm_amd", "libacvm_amd.so")]
    rows = []
    for f in files:
        for notes in code_objects(f):
            for blk in notes.split("- .agpr_count")[1:]:
                g = lambda k: (re.search(rf"\.{k}:\s+(\S+)", blk) or [None, "?"])[1]
                name = subprocess.run(["c++filt", g("name")], capture_output=True, text=True).stdout.strip()
                rows.append((re.sub(r"\(.*", "", name), g("vgpr_count"), g("sgpr_count"), g("private_segment_fixed_size"), g("group_segment_fixed_size")))
    print(f"{'kernel':90s} vgpr sgpr scratch lds")
    for r in sorted(set(rows)):
        print(f"{r[0][:90]:90s} {r[1]:>4s} {r[2]:>4s} {r[3]:>7s} {r[4]:>5s}")


if __name__ == "__main__":
    main()
