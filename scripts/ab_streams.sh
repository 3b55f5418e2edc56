cd $GRAFT_REPO_ROOT
B="python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-experiment"
show() { tail -1 $1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$2', d['ms_per_step'], 'host', d['host_issue_ms_per_step'], 'kernel', d['roofline']['ms_per_launch'])"; }
$B > gpurun_out/ab_default.txt 2>&1; show gpurun_out/ab_default.txt default
PCGAN_SPLIT=bf16 $B > gpurun_out/ab_bf16.txt 2>&1; show gpurun_out/ab_bf16.txt split=bf16
PCGAN_SIDE_STREAM=0 $B > gpurun_out/ab_noside.txt 2>&1; show gpurun_out/ab_noside.txt noside
PCGAN_BRANCH_STREAMS=0 $B > gpurun_out/ab_nobranch.txt 2>&1; show gpurun_out/ab_nobranch.txt nobranch
PCGAN_SIDE_STREAM=0 PCGAN_BRANCH_STREAMS=0 $B > gpurun_out/ab_single.txt 2>&1; show gpurun_out/ab_single.txt single
bash scripts/profile_single_stream.sh hs_ss > /dev/null 2>&1
python scripts/group_stats.py gpurun_out/hs_ss/kernel_stats_fp32.csv 13 25
