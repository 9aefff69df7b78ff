#!/bin/bash
# Code-object notes (VGPRs, SGPRs, spills, LDS, scratch) of every kernel in a hipcc object or shared library.
# usage: bash profiles/notes.sh ada-ray-tracer_amd/build/art_kernels.hip.o [name filter]
O=$1; F=${2:-.}
T=$(mktemp -d -p "${TMPDIR:-/tmp}")
B=/opt/rocm/lib/llvm/bin
$B/llvm-objcopy -O binary --only-section=.hip_fatbin $O $T/fat.bin      # the device code sits in the host object's .hip_fatbin section as an offload bundle
$B/clang-offload-bundler --type=o --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --input=$T/fat.bin --output=$T/dev.co --unbundle
$B/llvm-readelf --notes $T/dev.co | python3 -c '
import sys, re, subprocess
txt = sys.stdin.read()
rows = []
for blk in re.split(r"\n\s*- \.agpr_count", txt)[1:]:
    g = lambda k: (re.search(r"\." + k + r":\s*(\S+)", blk) or [None, "?"])[1]
    rows.append((g("name"), g("vgpr_count"), g("sgpr_count"), g("vgpr_spill_count"), g("sgpr_spill_count"), g("group_segment_fixed_size"), g("private_segment_fixed_size")))
names = subprocess.run(["c++filt"], input="\n".join(r[0] for r in rows), capture_output=True, text=True).stdout.split("\n")
for r, n in zip(rows, names):
    n = re.sub(r"\(.*", "", n).replace("void ", "").replace("art::", "")
    print("%-44s vgpr %3s sgpr %3s spilled_vgpr %3s spilled_sgpr %3s lds %6s scratch %4s" % ((n[:44],) + r[1:]))
' | grep -E "$F"
rm -rf $T
