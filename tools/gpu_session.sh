#!/bin/bash
# One GPU-box session (run through gpurun), assembled from named steps; everything lands under gpurun_out/session/.
#   usage: gpu_session.sh step [step ...]
#   tests[:<pytest -k expression>]   pytest -m gpu (the whole suite, or the selected tests) with durations
#   bench                            python bench.py --steps 20 --warmup 5   (what the driver runs; secondaries + live counter traffic)
#   stats                            rocprofv3 --kernel-trace --stats of bench.py (headline only / secondaries only)
#   kpart-shapes | gram-first         round 5: SpMM partitioned vs row-owned across shapes; kernels of a first dense gram call
#   pmc-kpart[:variant]              the same for tools/gpu_kpart.py (column-partitioned SpMM; variant e.g. 128:8 or off)
#   pmc-spmm | pmc-gram | pmc-spgemm separate rocprofv3 --pmc passes for the headline SpMM kernel / the dense gram kernel / every SpGEMM kernel of the
#                                    literal and the uniform configs[2] (tools/pmc_kernels.py)
#   probes                           tools/probes: spmm_gather_probe (+ counters for H = 16384), lds_atomic_probe, host_register_probe
#   phases                           SpGEMM phase trace (uniform + literal) and kernel stats of the uniform call
#   api                              host-array call breakdown, stager on / off
#   det                              cost of option deterministic
#   dryrun8                          bench.py --gpus 8 as eight gloo ranks on this one GPU (control flow only)
cd $GRAFT_REPO_ROOT; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/session; mkdir -p $O
pmc() {  # $1 = output tag, rest = command; four counter groups, one pass each
  local tag=$1; shift; local i=0
  for grp in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_BUSY_CYCLES"; do
    i=$((i+1)); ( cd /tmp && TMPDIR=/tmp timeout 600 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $O/$tag/p$i -o p -- "$@" > $O/$tag.p$i.log 2>&1 )
  done
  python tools/pmc_kernels.py $O/$tag | grep "mi::" > $O/$tag.jsonl; rm -rf $O/$tag; cut -c1-600 $O/$tag.jsonl | head -8
}
for step in "$@"; do
  echo "#### $step"
  case $step in
    tests*) k=${step#tests}; k=${k#:}
      ( time timeout 3000 python -m pytest tests -m gpu -q -x -p no:cacheprovider --durations=10 ${k:+-k "$k"} ) > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -16 $O/pytest_gpu.log ;;
    bench)
      ( time timeout 1800 python bench.py --steps 20 --warmup 5 ) > $O/bench.log 2> $O/bench.err; echo "bench rc=$?"; grep '^{' $O/bench.log | tail -1 > $O/bench_line.json
      python - <<'PY'
import json,os
d=json.load(open(os.environ["GRAFT_REPO_ROOT"]+"/gpurun_out/session/bench_line.json"))
print(json.dumps({k:d[k] for k in ("value","ms_per_step","roofline")})[:900])
for k,v in d.get("secondary",{}).items():
    if k=="gemm_dense":
        print(k, {t:(v[t].get("value"), v[t].get("roofline",{}).get("frac")) for t in ("f32","f64") if t in v}); continue
    rf=v.get("roofline") or {}
    print(k, {kk:v.get(kk) for kk in ("ms","ms_per_step","value","error") if kk in v}, {kk:rf.get(kk) for kk in ("frac","traffic","traffic_over_algorithmic")}, v.get("gpu_on_cpu_sample_shape"))
PY
      tail -3 $O/bench.err ;;
    stats)
      ( cd /tmp && TMPDIR=/tmp timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/st1 -o s -- python $R/bench.py --steps 20 --warmup 5 --no-cpu --no-secondary --no-pmc > $O/stats_headline.log 2>&1 )
      cp $(find $O/st1 -name "*kernel_stats.csv" | head -1) $O/bench_kernel_stats.csv; rm -rf $O/st1; head -4 $O/bench_kernel_stats.csv | cut -c1-160
      ( cd /tmp && TMPDIR=/tmp timeout 1500 rocprofv3 --kernel-trace --stats --output-format csv -d $O/st2 -o s -- python $R/bench.py --steps 3 --warmup 1 --no-cpu --no-pmc --secondary spgemm,spgemm_rmat,gram > $O/stats_secondary.log 2>&1 )
      cp $(find $O/st2 -name "*kernel_stats.csv" | head -1) $O/secondary_kernel_stats.csv; rm -rf $O/st2; head -8 $O/secondary_kernel_stats.csv | cut -c1-160 ;;
    pmc-spmm) pmc pmc_spmm python $R/tools/spmm_sweep.py --launches 5 --variants 0:8192:256 --adopt-tags ;;
    pmc-kpart*) v=${step#pmc-kpart}; v=${v#:}; pmc pmc_kpart python $R/tools/gpu_kpart.py --variants ${v:-128:8} --launches 5 --warm 6 ;;
    kpart-shapes)  # the column-partitioned SpMM against the row-owned kernel across widths, dtypes and sizes
      { for a in "--ncols 128" "--ncols 256" "--ncols 64" "--dtype f64 --ncols 128" "--scale 22 --launches 5" "--workload uniform"; do echo "## $a"; timeout 500 python tools/gpu_kpart.py $a --variants "off,default" 2>&1 | grep "variant\|workload" | cut -c1-330; done; } > $O/spmm_kpart_shapes.log; cat $O/spmm_kpart_shapes.log | cut -c1-140 ;;
    gram-first)  # kernels of the FIRST dense gram on a fresh handle (transpose, tables) at the literal configs[3]
      ( cd /tmp && TMPDIR=/tmp timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/gf -o s -- python $R/tools/bench_ops.py gram --dense --cols 262144 --rows-log2 22 --reps 1 > $O/gf.log 2>&1 ); python tools/kstats.py $O/gf > $O/gram_first_call_kernel_stats.log; rm -rf $O/gf; head -12 $O/gram_first_call_kernel_stats.log ;;
    pmc-gram) pmc pmc_gram python $R/tools/bench_ops.py gram --dense --cols 262144 --rows-log2 22 --reps 1 ;;
    pmc-spgemm) pmc pmc_spgemm_literal python $R/tools/bench_ops.py spgemm --kind rmat --scale 20 --per-row 16 --no-order --reps 1
                pmc pmc_spgemm_uniform python $R/tools/bench_ops.py spgemm --no-order --reps 2 ;;
    probes)
      tools/probes/spmm_gather_probe > $O/spmm_gather_probe.log 2>&1; tail -3 $O/spmm_gather_probe.log
      ( cd /tmp && TMPDIR=/tmp timeout 600 rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum --output-format csv -d $O/gp -o p -- $R/tools/probes/spmm_gather_probe H=16384 > $O/spmm_gather_probe_pmc.log 2>&1 )
      python tools/probe_pmc_join.py $O/spmm_gather_probe_pmc.log $O/gp > $O/spmm_gather_probe_pmc_joined.log; rm -rf $O/gp
      tools/probes/lds_atomic_probe | tee $O/lds_atomic_probe.log; python tools/probes/host_register_probe.py | tee $O/host_register_probe.log ;;
    phases)
      for kind in "" "--kind rmat --scale 20 --per-row 16"; do MI_BENCH_OPTS=trace_phases=1 timeout 300 python tools/bench_ops.py spgemm $kind --no-order --reps 1 2>&1 | grep "mi_sparse spgemm" | tail -6; done | tee $O/spgemm_phases.log
      ( cd /tmp && TMPDIR=/tmp timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/ph -o u -- python $R/tools/bench_ops.py spgemm --no-order --reps 5 > $O/ph.log 2>&1 ); cp $(find $O/ph -name "*kernel_stats.csv" | head -1) $O/spgemm_uniform_kernel_stats.csv; rm -rf $O/ph ;;
    api) timeout 600 python tools/gpu_api_overhead.py > $O/api_overhead.log 2>&1; grep "####\|== \|mi_sparse_s_mm \|create_csr\|python" $O/api_overhead.log | head -40 ;;
    det) for det in 0 1; do echo "== deterministic=$det"; for a in "spgemm --no-order --reps 2" "spgemm --kind rmat --scale 16 --per-row 16 --no-order --reps 2" "gram --dense --cols 16384 --rows-log2 20 --reps 2"; do MI_BENCH_OPTS=deterministic=$det timeout 600 python tools/bench_ops.py $a 2>&1 | tail -1 | cut -c1-220; done; done | tee $O/deterministic_cost.log ;;
    dryrun8)
      ( export BENCH_DIST_BACKEND=gloo BENCH_CFG5_SCALE=14 HSA_ENABLE_IPC_MODE_LEGACY=0; time timeout 1500 python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port 29541 bench.py --gpus 8 --steps 2 --warmup 1 --scale 16 --no-cpu --no-pmc ) > $O/bench_8rank_dryrun.log 2> $O/bench_8rank_dryrun.err; echo "8-rank rc=$?"
      grep '^{' $O/bench_8rank_dryrun.log | tail -1 > $O/bench_8rank_line.json; cut -c1-600 $O/bench_8rank_line.json ;;
    *) echo "unknown step $step" ;;
  esac
done
