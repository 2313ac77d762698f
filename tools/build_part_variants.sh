#!/bin/bash
# Developer tool: build libmi_sparse variants with different big-row SpGEMM tuning constants (run here, they travel to the GPU box).
set -e
cd "$(dirname "$0")/../sparse_dot_amd/csrc"
mkdir -p build/var
SPECS="xr0:-DMI_PART_XCD_RUN=0 xr8:-DMI_PART_XCD_RUN=8 xr128:-DMI_PART_XCD_RUN=128 xr32g4:-DMI_PART_GROUP=4"
for spec in $SPECS; do
  tag=${spec%%:*}; def=$(echo ${spec#*:} | tr '@' ' ')
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics $def -c spgemm.hip -o build/var/spgemm_$tag.o &
done
wait
for spec in $SPECS; do
  tag=${spec%%:*}
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o build/var/libmi_sparse_$tag.so build/runtime.o build/handle.o build/spmm.o build/var/spgemm_$tag.o build/gram.o build/dense.o build/bsr.o
done
ls build/var/*.so
