#!/bin/bash
# Build libmi_sparse variants of ONE source file from -D specs (run here; the .so files travel to the GPU box under
# sparse_dot_amd/csrc/build/var/ and are selected there with MI_SPARSE_RT -- tools/gpu_ab.sh).  How the knock-outs and tuning A/Bs of
# rounds 2-4 were run.       usage: build_variants.sh gram|spgemm|spmm  tag:-DFLAG[@-DFLAG2] ...      or   tag:@<other source file>
set -e
cd "$(dirname "$0")/../sparse_dot_amd/csrc"; src=$1; shift; mkdir -p build/var
for spec in "$@"; do
  tag=${spec%%:*}; def=$(echo ${spec#*:} | tr '@' ' '); in=$src.hip
  case "$def" in " "*) in=${def# }; def="";; esac   # tag:@file = compile another copy of the source (e.g. a git show of the previous version)
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics -I. $def -x hip -c $in -o build/var/${src}_$tag.o &
done
wait
for spec in "$@"; do
  tag=${spec%%:*}; objs=""
  for f in runtime handle spmm spgemm gram dense bsr; do if [ $f = $src ]; then objs="$objs build/var/${src}_$tag.o"; else objs="$objs build/$f.o"; fi; done
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o build/var/libmi_sparse_$tag.so $objs
done
ls build/var/*.so
