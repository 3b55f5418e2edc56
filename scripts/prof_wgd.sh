cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/wgd -- python3 $GRAFT_REPO_ROOT/scripts/time_wgd.py > /tmp/wgd.log 2>&1; tail -9 /tmp/wgd.log; python3 - <<PY
import csv,glob
f=glob.glob("/tmp/wgd/**/*kernel_stats.csv",recursive=True)[0]
for r in csv.DictReader(open(f)):
    if "wgd" in r["Name"] or "hsplit_wgrad" in r["Name"] or "wgrad_reduce" in r["Name"]:
        print(r["Name"][:70], r["Calls"], r["AverageNs"])
PY
