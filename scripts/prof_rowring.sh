cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/rr && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/rr -- python3 $GRAFT_REPO_ROOT/scripts/time_rowring.py "$@" > /tmp/rr.log 2>&1; tail -4 /tmp/rr.log; python3 - <<PY
import csv,glob
f=glob.glob("/tmp/rr/**/*kernel_stats.csv",recursive=True)[0]
for r in csv.DictReader(open(f)):
    if "rowring" in r["Name"] or "wgd_reduce" in r["Name"] or "hsplit_wgrad" in r["Name"] or "wgrad_reduce" in r["Name"]:
        print(r["Name"][:70], r["Calls"], r["AverageNs"])
PY
