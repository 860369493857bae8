#!/usr/bin/env python3
"""VGPR / SGPR / spill / LDS figures of every kernel in the built library (read from the gfx950 code object's metadata notes).
usage: tools/kernel_resources.py [lib.so] [name-filter]"""
import re, subprocess, sys, os, tempfile
lib = sys.argv[1] if len(sys.argv) > 1 and sys.argv[1].endswith(".so") else os.path.join(os.path.dirname(__file__), "..", "ipopt_amd", "lib", "libmi355x_kkt.so")
flt = sys.argv[-1] if len(sys.argv) > 1 and not sys.argv[-1].endswith(".so") else ""
llvm = "/opt/rocm/lib/llvm/bin"
with tempfile.TemporaryDirectory() as td:
    # the fat binary sits in .hip_fatbin of the shared library
    subprocess.check_call([f"{llvm}/llvm-objcopy", "-O", "binary", "--only-section=.hip_fatbin", lib, f"{td}/fat"])
    subprocess.check_call([f"{llvm}/clang-offload-bundler", "--unbundle", "--type=o", f"--input={td}/fat", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", f"--output={td}/co"])
    txt = subprocess.check_output([f"{llvm}/llvm-readelf", "--notes", f"{td}/co"], text=True)
rows = []
for blk in txt.split("- .agpr_count:")[1:]:
    g = lambda k: (re.search(rf"\.{k}:\s+(\S+)", blk) or [None, "?"])[1]
    name = subprocess.run(["c++filt", g("name")], capture_output=True, text=True).stdout.strip()
    name = re.sub(r"\(mi355x::DevView.*", "", name).replace("mi355x::", "").replace("void ", "")
    rows.append((name, g("vgpr_count"), g("vgpr_spill_count"), g("sgpr_count"), g("sgpr_spill_count"), g("group_segment_fixed_size"), g("private_segment_fixed_size")))
print(f"{'kernel':60s} vgpr vspill sgpr sspill   lds scratch")
for r in sorted(rows):
    if flt in r[0]:
        print(f"{r[0][:60]:60s} {r[1]:>4s} {r[2]:>6s} {r[3]:>4s} {r[4]:>6s} {r[5]:>5s} {r[6]:>7s}")
