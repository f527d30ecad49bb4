#!/bin/bash
# ISA of ONE kernel instantiation without the 4-minute build of every variant: the device code of a kernels file up to its
# launchers + one explicit instantiation -> hipcc -c -> llvm-objdump.   Usage:
#   tools/isa_one.sh 'part_scatter_wv<2, 1, 0, true, 0, 3, 0, 0>' [vxh_kernels.hip] [extra hipcc flags]   -> /tmp/isa/one.s (+ one.notes)
set -e
INST="$1"; SRC="${2:-vxh_kernels.hip}"; shift; shift || true
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
OUT=/tmp/isa; mkdir -p $OUT
CS="$ROOT/vaex_amd/csrc"
END=$(grep -n '^} // namespace' "$CS/$SRC" | head -1 | cut -d: -f1)
head -n $((END-1)) "$CS/$SRC" > $OUT/one.hip
echo "template __global__ void ${INST}(const $(echo "$INST" | grep -q '^gb_' && echo GbArgs || (echo "$INST" | grep -q 'count_lds\|bin_kernel' && echo BinArgs || echo PartArgs)));" >> $OUT/one.hip
echo "} // namespace" >> $OUT/one.hip
echo "void *vxh_isa_ref() { return (void *)&${INST}; }" >> $OUT/one.hip
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -munsafe-fp-atomics -fPIC -Wno-unused-function -I"$CS" -I"$ROOT/include" "$@" -c $OUT/one.hip -o $OUT/one.o
export PATH=$PATH:/opt/rocm/lib/llvm/bin
objcopy -O binary --only-section=.hip_fatbin $OUT/one.o $OUT/one.fat
clang-offload-bundler --unbundle --type=o --input=$OUT/one.fat --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --output=$OUT/one.co
llvm-objdump -d $OUT/one.co | sed 's/ *\/\/ [0-9A-F]*:.*//' > $OUT/one.s
llvm-readelf --notes $OUT/one.co > $OUT/one.notes
python3 "$ROOT/tools/kregs.py" $OUT/one.notes
echo "lines: $(wc -l < $OUT/one.s)  scratch ops: $(grep -c scratch_ $OUT/one.s || true)  vmcnt(0): $(grep -c 'vmcnt(0)' $OUT/one.s || true)"
