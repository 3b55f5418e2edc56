"""CPU: the bench.py contract -- flags, and the shape of the JSON line (checked on the latest line committed under
profiles/, which bench.py printed on an MI355X)."""
import glob
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_flags_and_defaults():
    out = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--help'], capture_output=True, text=True, cwd=ROOT).stdout
    for flag in ('--gpus', '--steps', '--warmup'):
        assert flag in out
    import bench
    assert bench.PER_GPU_BATCH == 32 and bench.SIZE == 128 and bench.GFLOP_PER_IMG_FULL == 183.95
    assert bench.RES_CONV_FLOP == 2.0 * 32 * 32 * 32 * 256 * 256 * 9


def test_committed_line_has_the_contract_fields():
    lines = sorted(glob.glob(os.path.join(ROOT, 'profiles', 'r01_bench_n1_v*.json')), key=lambda p: int(p.split('_v')[-1].split('.')[0]))
    line = json.loads(open(lines[-1]).read())
    for key in ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'higher_is_better', 'scaling', 'vs_baseline',
                'dtype', 'data', 'config', 'roofline', 'cpu_baseline'):
        assert key in line, key
    assert line['unit'] == 'images/sec' and line['higher_is_better'] is True and line['scaling'] == 'weak' and line['vs_baseline'] is None
    assert line['dtype'] == 'f32' and line['data'] == 'synthetic' and 'workload' in line['config'] and 'model' not in line['config']
    assert abs(line['value'] - 32 * line['n_gpus'] / line['ms_per_step'] * 1e3) < 0.01 * line['value']
    r = line['roofline']
    assert r['bound'] == 'mfma' and r['unit'] == 'TFLOP/s' and abs(r['frac'] - r['achieved'] / r['peak']) < 1e-3 and r['traffic'] > 0
    assert abs(r['achieved'] - r['flop_per_launch'] / (r['ms_per_launch'] * 1e-3) / 1e12) < 0.05 and r['launches_timed'] == 36 * line['steps']
    if 'experiment_bf16x6' in line:      # labelled as not the default path; the headline stays the fp32 MFMA measurement
        assert line['experiment_bf16x6']['default'] is False and line['experiment_bf16x6']['value'] > 0
    c = line['cpu_baseline']
    assert c['kind'] in ('port', 'reference') and c['cores'] >= 1 and c['value'] > 0 and c['sample']


def test_round4_line_says_exactly_what_ran():
    """the line names its arithmetic, uploads its batches inside the timed region, reports the CPU baseline at the
    best thread count of a sweep, and carries the counter-derived matrix-pipe busy share per residual kernel.  Checked on the line
    committed this round (profiles/r04_bench_n1.json, printed by bench.py on an MI355X)."""
    path = os.path.join(ROOT, 'profiles', 'r04_bench_n1.json')
    assert os.path.exists(path), 'commit this round\'s bench line as profiles/r04_bench_n1.json'
    line = json.loads(open(path).read())
    assert line['dtype'] == 'f32' and 'fp16 pieces' in line['arithmetic'] and '3 piece products' in line['arithmetic']
    assert 'inside the timed region' in line['input']
    c = line['cpu_baseline']
    assert c['kind'] == 'port' and str(c['cores']) in c['threads_swept'] and len(c['threads_swept']) >= 2
    assert c['value'] >= max(c['threads_swept'].values()) * 0.6      # the reported value is the winner's, not an oversubscribed default
    assert c['host_cpus']['usable'] >= 1
    r = line['roofline']
    assert set(r['kernels']) == {'res_fwd', 'res_dgrad', 'res_wgrad'}
    for k, e in r['kernels'].items():
        assert 'mfma_busy' in e and 'frac' in e and 'frac_alone' in e and e['traffic'] is not None, k
        assert e['mfma_busy'] is None or 0.0 < e['mfma_busy'] <= 1.0
    assert 'route_bf16x6' in line and 'route_fp32_mfma' in line
    assert line['host_issue_ms_per_step'] > 0


def test_cpu_budget_is_cgroup_aware():
    import bench
    n, why = bench.host_cpu_budget()
    assert 1 <= n <= (os.cpu_count() or 1) and 'affinity' in why
