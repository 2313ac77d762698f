#!/bin/bash
# Developer tool: build libmi_sparse variants with different big-row SpGEMM tuning constants (run here, they travel to the GPU box).
set -e
cd "$(dirname "$0")/../sparse_dot_amd/csrc"
mkdir -p build/var
for spec in "sp2:-DMI_SLICE_PASSES=2" "sp4:-DMI_SLICE_PASSES=4" "sp16:-DMI_SLICE_PASSES=16"; do
  tag=${spec%%:*}; def=${spec#*:}
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics $def -c spgemm.hip -o build/var/spgemm_$tag.o &
done
wait
for tag in sp2 sp4 sp16; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o build/var/libmi_sparse_$tag.so build/runtime.o build/handle.o build/spmm.o build/var/spgemm_$tag.o build/gram.o build/dense.o
done
ls -la build/var/*.so
