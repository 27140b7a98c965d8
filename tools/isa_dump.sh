#!/bin/bash
# gfx950 ISA + kernel metadata of one translation unit of the built library (development aid).
# usage: tools/isa_dump.sh topk [obj-dir]   ->  /tmp/clipmi_isa/topk.s, /tmp/clipmi_isa/topk.notes
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
TU=${1:-topk}
OBJ=${2:-$ROOT/cli-p_amd/csrc/_obj}
OUT=/tmp/clipmi_isa
LLVM=/opt/rocm/lib/llvm/bin
mkdir -p "$OUT"
cp "$OBJ/$TU.o" "$OUT/$TU.o"
(cd "$OUT" && "$LLVM/llvm-objdump" --offloading "$TU.o" > /dev/null 2>&1)
mv "$OUT/$TU.o.0.hipv4-amdgcn-amd-amdhsa--gfx950" "$OUT/$TU.co"
"$LLVM/llvm-objdump" -d "$OUT/$TU.co" > "$OUT/$TU.s"
"$LLVM/llvm-readelf" --notes "$OUT/$TU.co" > "$OUT/$TU.notes"
echo "$OUT/$TU.s $OUT/$TU.notes"
