#!/bin/bash
# Developer tool: compile the kernels' source for HOST threads (see hip_emu.hpp).  Output goes to
# /tmp so it can never be picked up by the package or shipped to the GPU box.
set -e
HERE="$(cd "$(dirname "$0")" && pwd)"
SRC="$HERE/../../sparse_dot_amd/csrc"
OUT=${1:-/tmp/libmi_sparse_emu.so}
mkdir -p /tmp/mi_emu_build
for f in runtime handle spmm spgemm gram dense bsr; do
  rm -f /tmp/mi_emu_build/$f.o; g++ -x c++ -std=c++17 -O1 -g -fPIC -pthread -DMI_HIP_EMU -Wno-psabi -I"$HERE" -I"$SRC" -c "$SRC/$f.hip" -o /tmp/mi_emu_build/$f.o &
done
wait; for f in runtime handle spmm spgemm gram dense bsr; do [ -f /tmp/mi_emu_build/$f.o ] || { echo "error: $f failed"; exit 1; }; done
g++ -shared -pthread -o "$OUT" /tmp/mi_emu_build/*.o
echo "built $OUT"
