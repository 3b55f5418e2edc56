"""group a rocprofv3 kernel_stats.csv into the step's kernel families: ms per step.  usage: group_stats.py <csv> <steps> [top]"""
import collections, csv, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
steps = float(sys.argv[2])
top = int(sys.argv[3]) if len(sys.argv) > 3 else 0
groups = collections.OrderedDict([
    ('residual convs (bsplit main)', r'bsplit_conv_fwd_kernel|bsplit_halo_kernel|hsplit_wgrad_kernel|rowring_wgrad'), ('residual wgrad aux (pad / pack / reduce)', r'bsplit_(pad|pack_dy|wgrad_reduce)|wgd_reduce_kernel'),
    ('bsplit weight packs, absmax', r'bsplit_pack|absmax'), ('other MFMA convs fwd+dgrad (igemm2 / igemm)', r'igemm2?_kernel|hgemm_kernel'),
    ('other MFMA wgrad (wgrad2 / wgrad + reduce)', r'::wgrad2?_kernel|::wgrad_reduce|\d\dwgrad2?_kernelI'), ('<=4-channel convs (smallm, thin_conv)', r'smallm|thin_conv|thin_pack'),
    ('split-K reduce / repack / fold', r'splitk|repack|reflect_fold|transpose4|pack_strip'), ('instance norm', r'instnorm|in_running'),
    ('batch norm / plane stats', r'bn_|norm_|plane_stats'), ('bias sums', r'sum_over_n|plane_sum'),
    ('pointwise / cast', r'act_|add_kernel|scale_kernel|concat_z|channel_scale'), ('pool / resize', r'pool|bilinear'),
    ('loss / adam', r'loss|adam|incr'), ('torch glue', r'at::native')])
acc = collections.OrderedDict((k, [0.0, 0]) for k in groups)
acc['other'] = [0.0, 0]
tot = 0.0
for r in rows:
    ns, calls = float(r['TotalDurationNs']), int(r['Calls'])
    tot += ns
    for k, pat in groups.items():
        if re.search(pat, r['Name']):
            acc[k][0] += ns; acc[k][1] += calls
            break
    else:
        acc['other'][0] += ns; acc['other'][1] += calls
for k, (ns, calls) in acc.items():
    print('%-50s %7.2f ms/step  %6.0f launches/step' % (k, ns / 1e6 / steps, calls / steps))
print('%-50s %7.2f ms/step' % ('total kernel time', tot / 1e6 / steps))
for r in sorted(rows, key=lambda r: -float(r['TotalDurationNs']))[:top]:
    print('  %-90s %5d calls  avg %8.1f us  %6.2f ms/step' % (r['Name'][:90], int(r['Calls']), float(r['AverageNs']) / 1e3, float(r['TotalDurationNs']) / 1e6 / steps))
