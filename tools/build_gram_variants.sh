#!/bin/bash
# Developer tool: libmi_sparse variants of the dense gram kernel (experiment hooks; run here, they travel to the GPU box).
set -e
cd "$(dirname "$0")/../sparse_dot_amd/csrc"
mkdir -p build/var
SPECS="${SPECS:-gr8:-DMI_GRAM_R=8 gr2:-DMI_GRAM_R=2 sub16:-DMI_GRAM_SUB=16 sub4:-DMI_GRAM_SUB=4 scalarst:-DMI_GRAM_NT_STORE=0}"
for spec in $SPECS; do
  tag=${spec%%:*}; def=$(echo ${spec#*:} | tr '@' ' ')
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics $def -c gram.hip -o build/var/gram_$tag.o &
done
wait
for spec in $SPECS; do
  tag=${spec%%:*}
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o build/var/libmi_sparse_$tag.so build/runtime.o build/handle.o build/spmm.o build/spgemm.o build/var/gram_$tag.o build/dense.o build/bsr.o
done
ls build/var/*.so
