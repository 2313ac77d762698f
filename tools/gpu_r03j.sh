# Round-3 session J: host-array path -- where the time goes now; host memory settings of the box
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r03j; O=$GRAFT_REPO_ROOT/gpurun_out/r03j
( cat /sys/kernel/mm/transparent_hugepage/enabled /sys/kernel/mm/transparent_hugepage/defrag 2>&1; nproc; grep -i "hugepages_total\|MemTotal\|MemFree" /proc/meminfo; uname -r ) | tee $O/host_settings.log
ONLY_STAGED=1 timeout 600 python tools/gpu_api_overhead.py 2>&1 | grep -v amdgpu.ids | tee $O/api_overhead.log | head -40
