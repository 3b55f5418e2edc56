"""profiles/<tag>_residual_kernel_traffic.json from the FETCH_SIZE / WRITE_SIZE passes of scripts/profile_round.sh (rocprofv3 --pmc over
scripts/run_res_conv.py): HBM-side bytes per launch of the three residual-convolution kernels, the weight gradient INCLUDING its helper
launches (padded copy where there still is one, reduce).  usage: make_traffic_json.py <gpurun_out/tag> <out.json> <route>"""
import json, re, sys
out_dir, out_json, route = sys.argv[1], sys.argv[2], sys.argv[3]


def means(counter):
    """kernel name prefix -> mean counter value per launch (KiB), from pmc_<counter>.txt"""
    res, cur = {}, None
    for line in open('%s/pmc_%s.txt' % (out_dir, counter)):
        m = re.match(r'^(\S.*?)\s+mean duration', line)
        if m:
            cur = m.group(1)
        m = re.match(r'^\s+%s\s+n=(\d+) mean=([0-9.e+]+)' % counter, line)
        if m and cur:
            res[cur] = float(m.group(2))
    return res


f, w = means('FETCH_SIZE'), means('WRITE_SIZE')


def pick(d, pat):
    return {k: v for k, v in d.items() if re.search(pat, k)}


alg = 32 * 256 * 32 * 32 * 4 * 2 + 256 * 256 * 9 * 4          # one activation tensor in, one out, the weights (as fp32 bytes)
alg_w = 32 * 256 * 32 * 32 * 4 * 2 + 256 * 256 * 9 * 4        # weight gradient: x and dy in, dw out
spec = {'fwd': r'bsplit_halo_kernel<0, 2, float, 32>', 'dgrad': r'bsplit_halo_kernel<1, 2, float, 32>',
        'wgrad': r'hsplit_wgrad_kernel<256, 1, float, 1>|bsplit_wgrad_reduce_kernel|bsplit_pad_wave_kernel'}
res = {'route': route, 'shape': '256->256 3x3 reflect @32x32, bs32', 'unit_note': 'FETCH_SIZE / WRITE_SIZE raw, KiB -> bytes x 1024 (profiles/README.md: the gfx950 x2 rule '
       'for 16-byte streaming reads is NOT applied: uncalibrated for these access patterns; ratios between rounds are unaffected)',
       'source': 'profiles/r03_counters_residual_convs.txt: rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) --kernel-trace -- python3 scripts/run_res_conv.py 5'}
for k, pat in spec.items():
    fk, wk = pick(f, pat), pick(w, pat)
    res[k] = {'fetch_bytes': int(sum(fk.values()) * 1024), 'write_bytes': int(sum(wk.values()) * 1024),
              'algorithmic_bytes': alg_w if k == 'wgrad' else alg, 'kernels': sorted(fk)}
json.dump(res, open(out_json, 'w'), indent=1)
print(json.dumps(res, indent=1))
