#!/usr/bin/env python3
"""Register / spill / LDS figures of the kernels of a code object's metadata notes (llvm-readelf --notes output).
Usage: tools/kregs.py <notes.txt> [substring ...]   — build the notes with:
  objcopy -O binary --only-section=.hip_fatbin vaex_amd/lib/vxh_kernels.o fat.bin
  clang-offload-bundler --unbundle --type=o --input=fat.bin --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --output=k.co
  llvm-readelf --notes k.co > notes.txt"""
import re
import sys
txt = open(sys.argv[1]).read()
pats = sys.argv[2:]
for b in txt.split("- .agpr_count")[1:]:
    m = re.search(r"\.name:\s+(\S+)", b)
    if not m or (pats and not any(p in m.group(1) for p in pats)):
        continue
    g = lambda k: re.search(r"\." + k + r":\s+(\d+)", b).group(1)
    print(f"{m.group(1)[:70]:<70} vgpr {g('vgpr_count'):>3} sgpr {g('sgpr_count'):>3} vspill {g('vgpr_spill_count'):>3} sspill {g('sgpr_spill_count'):>3} scratch {g('private_segment_fixed_size'):>5}")
