#!/bin/bash
# Register / spill / kernarg / LDS figures of every kernel in the built libcrt_hip.so (gfx950 code object): llvm-readelf --notes.
set -e
LIB=${1:-$(dirname "$0")/../course-assignment-danielhalachev_amd/libcrt_hip.so}
LLVM=/opt/rocm/lib/llvm/bin
T=$(mktemp -d)
# the fat binary sits in .hip_fatbin; the bundle's gfx950 entry is an ELF
$LLVM/llvm-objcopy -O binary --only-section=.hip_fatbin "$LIB" "$T/fat.bin"
$LLVM/clang-offload-bundler --type=o --input="$T/fat.bin" --unbundle --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --output="$T/dev.co" 2>/dev/null || \
$LLVM/clang-offload-bundler --type=o --input="$T/fat.bin" --unbundle --targets=hip-amdgcn-amd-amdhsa--gfx950 --output="$T/dev.co"
$LLVM/llvm-readelf --notes "$T/dev.co" | python3 -c '
import re, sys
txt = sys.stdin.read()
for blk in txt.split("- .agpr_count")[1:]:
    g = lambda k: (re.search(r"\.%s:\s+(\S+)" % k, blk) or [None, "?"])[1]
    name = g("name")
    import subprocess
    print("%-70s vgpr %4s sgpr %4s vspill %3s sspill %3s kernarg %5s lds %6s" % (name[:70], g("vgpr_count"), g("sgpr_count"), g("vgpr_spill_count"), g("sgpr_spill_count"), g("kernarg_segment_size"), g("group_segment_fixed_size")))
' | while read -r line; do n=$(echo "$line" | awk '{print $1}'); d=$(c++filt "$n" | sed 's/(anonymous namespace):://; s/((anonymous namespace)::KernelArgs.*//' | cut -c1-60); echo "$d ${line#* }"; done
rm -rf "$T"
