cd $GRAFT_REPO_ROOT
B="python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-experiment"
show() { tail -1 $1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$2', d['ms_per_step'], 'host', d['host_issue_ms_per_step'], 'kernel', d['roofline']['ms_per_launch'])"; }
$B > gpurun_out/abh_default.txt 2>&1; show gpurun_out/abh_default.txt hgemm=1
PCGAN_HGEMM=0 $B > gpurun_out/abh_off.txt 2>&1; show gpurun_out/abh_off.txt hgemm=0
bash scripts/profile_single_stream.sh hs_ss > /dev/null 2>&1
python scripts/group_stats.py gpurun_out/hs_ss/kernel_stats_fp32.csv 13 40
