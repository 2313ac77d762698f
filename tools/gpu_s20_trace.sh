cd $GRAFT_REPO_ROOT
MI_TRACE=1 timeout 900 python - <<'PY'
import sys; sys.argv=["bench_ops.py","spgemm","--kind","rmat","--scale","20","--per-row","16","--reps","1","--no-order"]
import sparse_dot_amd as sda
sda.mi_set_option("trace_phases", 1)
sys.path.insert(0, "tools")
import runpy; runpy.run_path("tools/bench_ops.py", run_name="__main__")
PY
