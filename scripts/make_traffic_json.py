"""profiles/<tag>_residual_kernel_traffic.json from the FETCH_SIZE / WRITE_SIZE passes of scripts/profile_round.sh (rocprofv3 --pmc over
scripts/run_res_conv.py): HBM-side bytes per launch of the three residual-convolution kernels, the weight gradient INCLUDING its helper
launches (padded copy where there still is one, reduce).  usage: make_traffic_json.py <gpurun_out/tag> <out.json> <route>"""
import json, re, sys
out_dir, out_json, route = sys.argv[1], sys.argv[2], sys.argv[3]
tag = (sys.argv[4] if len(sys.argv) > 4 else 'r04')


def means(counter, file_tag=None):
    """kernel name prefix -> mean counter value per launch (KiB), from pmc_<counter>.txt"""
    res, cur = {}, None
    for line in open('%s/pmc_%s.txt' % (out_dir, file_tag or counter)):
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
        'wgrad': r'rowring_wgrad|wgd_reduce_kernel|hsplit_wgrad_kernel<256, 1, float, 1(, 0)?>|bsplit_wgrad_reduce_kernel|bsplit_pad_wave_kernel'}
# gfx950 correction (MI355X_MICROARCH.md, HBM section): FETCH_SIZE tallies the 128-byte requests of 16-byte-per-lane reads at 64 bytes.
# Calibrated in the SAME pass on a known byte count: absmax_kernel<float> reads the 32 x 256 x 32 x 32 fp32 tensor exactly once.
known = 32 * 256 * 32 * 32 * 4
cal = [v for k, v in f.items() if 'absmax_kernel<float>' in k]
fetch_scale = known / (cal[0] * 1024) if cal else 2.0
assert 1.9 < fetch_scale < 2.1 or 0.95 < fetch_scale < 1.05, fetch_scale
res = {'route': route, 'shape': '256->256 3x3 reflect @32x32, bs32',
       'unit_note': 'counters in KiB; fetch_bytes = FETCH_SIZE x 1024 x fetch_scale, where fetch_scale = (bytes absmax_kernel<float> reads: %d) / (its FETCH_SIZE '
                    'in the same pass) -- the gfx950 rule of MI355X_MICROARCH.md (128-byte requests of 16-byte-per-lane loads are tallied at 64 bytes), '
                    'calibrated rather than assumed; every kernel listed here loads 16 bytes per lane.  WRITE_SIZE x 1024 is exact for 16-byte stores; the '
                    'weight gradient\'s partial sums are 4-byte stores (row-ring form: 32 x 9 x 256 x 256 x 4 = 75.5 MB, read back by its reduce).  Infinity-Cache hits are '
                    'counted (these are L2 <-> fabric requests, an upper bound of HBM bytes)' % known,
       'fetch_scale': round(fetch_scale, 4), 'fetch_size_raw_kib_absmax': cal[0] if cal else None,
       'source': 'profiles/%s_counters_residual_convs.txt:' % tag + ' rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) --kernel-trace -- python3 scripts/run_res_conv.py 5'}
for k, pat in spec.items():
    fk, wk = pick(f, pat), pick(w, pat)
    res[k] = {'fetch_bytes': int(sum(fk.values()) * 1024 * fetch_scale), 'fetch_size_raw_kib': round(sum(fk.values()), 1), 'write_bytes': int(sum(wk.values()) * 1024),
              'algorithmic_bytes': alg_w if k == 'wgrad' else alg, 'kernels': sorted(fk)}
# share of the active cycles the matrix pipes are busy: SQ_VALU_MFMA_BUSY_CYCLES (summed over the chip's 1024 SIMDs) over
# GRBM_GUI_ACTIVE (summed over the 8 XCDs) x 128 SIMDs per XCD -- its own --pmc pass
try:
    MF = 'SQ_VALU_MFMA_BUSY_CYCLES_GRBM_GUI_ACTIVE'
    busy, act = means('SQ_VALU_MFMA_BUSY_CYCLES', MF), means('GRBM_GUI_ACTIVE', MF)
    main = {'fwd': spec['fwd'], 'dgrad': spec['dgrad'], 'wgrad': r'rowring_wgrad|hsplit_wgrad_kernel<256, 1, float, 1(, 0)?>'}
    res['mfma_busy'] = {}
    for k, pat in main.items():
        b, a = pick(busy, pat), pick(act, pat)
        if b and a:
            res['mfma_busy'][k] = round(sum(b.values()) / (sum(a.values()) * 128.0), 4)
    res['mfma_busy_note'] = ('SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE x 128): busy SIMD-cycles of the matrix pipes over active cycles x SIMDs per XCD '
                             '(both counters are sums over the chip / the 8 XCDs); the weight gradient\'s figure is its main kernel\'s')
except (OSError, KeyError, ZeroDivisionError) as e:
    res['mfma_busy'] = None
    res['mfma_busy_note'] = 'no matrix-pipe pass found: %r' % (e,)
json.dump(res, open(out_json, 'w'), indent=1)
print(json.dumps(res, indent=1))
