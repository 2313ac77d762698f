cd /tmp && export TMPDIR=/tmp
for spec in "u:spgemm --reps 2 --no-order" "g:gram --reps 2"; do
  tag=${spec%%:*}; a=${spec#*:}
  timeout 600 rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_ops_$tag -- python $GRAFT_REPO_ROOT/tools/bench_ops.py $a 2>&1 | tail -1 | cut -c1-300
done
cd $GRAFT_REPO_ROOT
python - <<'PY'
import csv, glob
for tag in ("u", "g"):
    f = glob.glob("gpurun_out/prof_ops_%s/*/*kernel_trace.csv" % tag)[0]
    rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
    # last occurrence of k_row_ub marks the start of the last SpGEMM
    idx = [i for i, r in enumerate(rows) if "k_row_ub" in r["Kernel_Name"]]
    i0 = idx[-1]
    t0 = int(rows[i0]["Start_Timestamp"])
    print("==", tag)
    out = []
    for r in rows[max(0, i0 - 12):]:
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        if (s - t0) / 1e6 > 400: break
        out.append("%9.3f %8.3f  %s" % ((s - t0) / 1e6, (e - s) / 1e6, r["Kernel_Name"][:90]))
    print("\n".join(out[:120]))
PY
rm -rf gpurun_out/prof_ops_u gpurun_out/prof_ops_g
