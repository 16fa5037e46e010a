#!/usr/bin/env python
"""Summarise a tools/profile_round.sh output directory into markdown."""
import collections
import csv
import glob
import json
import os
import sys


def short(name):
    name = name.replace('(anonymous namespace)::', '')
    return name.split('(')[0][:70]


def main(out):
    print('# rocprofv3 summary (%s)\n' % os.path.basename(out))
    for f in sorted(glob.glob(os.path.join(out, 'bench_*.json'))):
        try:
            j = json.loads(open(f).read().strip().splitlines()[-1])
        except Exception as e:  # noqa
            print('* %s: unreadable (%s)' % (os.path.basename(f), e))
            continue
        r = j.get('roofline') or {}
        print('* `%s`: value **%.4g %s**, %.3f ms/step, filt Hits@10 %.5f, filt MRR %.6f; dominant kernel %s: '
              '%.1f %s = %.1f%% of peak (%.4f ms/launch)' % (
                  os.path.basename(f), j['value'], j['unit'], j['ms_per_step'], j['filtered_hits_at_10'],
                  j['filtered_mrr'], r.get('kernel'), r.get('achieved', 0), r.get('unit'), 100 * r.get('frac', 0),
                  r.get('kernel_ms', 0)))
        print('  * roofline: %s' % json.dumps({k: v for k, v in r.items() if k not in ('package_power', 'note')}))
        if r.get('package_power'):
            print('  * package_power (dominant kernel back to back): %s; clock_settle: %s' % (json.dumps(r['package_power']), json.dumps(j.get('clock_settle'))))
        if j.get('workload_detail'):
            print('  * workload: %s' % json.dumps(j['workload_detail']))
        if j.get('cpu_baseline'):
            print('  * cpu_baseline: %s' % json.dumps(j['cpu_baseline']))
        if j.get('parity_full_split'):
            print('  * parity_full_split: %s' % json.dumps(j['parity_full_split']))
        if j.get('f32_mfma_only'):
            print('  * f32_mfma_only: %s' % json.dumps(j['f32_mfma_only']))
        if j.get('secondary'):
            print('  * secondary: %s' % json.dumps(j['secondary']))
    titles = {'trace': 'bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-full-parity (incl. the 500-step set-up training)',
              'trace_eval': 'bench.py --steps 20 --warmup 5 --only-timed (the evaluate() replays alone; trained weights: the '
                            'set-up training shows as score_fwd / bwd rows)',
              'trace_l1': 'bench.py --only-timed --workload transe_l1_fb15k237 --weights xavier (TransE-L1 evaluate() replays)'}
    subs = sorted(os.path.basename(d) for d in glob.glob(os.path.join(out, 'trace*')) if os.path.isdir(d))
    for sub in subs:
        title = titles.get(sub, 'bench.py --only-timed: %s (see tools/profile_round.sh)' % sub[len('trace_'):])
        stats = glob.glob(os.path.join(out, sub, '*kernel_stats.csv')) + glob.glob(os.path.join(out, sub, '*', '*kernel_stats.csv'))
        if stats:
            print('\n## kernel-trace --stats: %s\n' % title)
            print('| kernel | calls | total ms | avg us | % |')
            print('|---|---|---|---|---|')
            for r in list(csv.DictReader(open(stats[0])))[:(40 if sub in ('trace', 'trace_eval') else 16)]:
                print('| %s | %s | %.3f | %.2f | %s |' % (short(r['Name']), r['Calls'], float(r['TotalDurationNs']) / 1e6,
                                                          float(r['AverageNs']) / 1e3, r['Percentage']))
    print('\n## PMC counters, mean per launch of the dominant kernels\n')
    print('| counter | kernel | launches | mean per launch |')
    print('|---|---|---|---|')
    for d in sorted(glob.glob(os.path.join(out, 'pmc*'))):
        if not os.path.isdir(d):
            continue
        print('| **%s** | | | |' % os.path.basename(d))
        for f in glob.glob(os.path.join(d, '*counter_collection.csv')) + glob.glob(os.path.join(d, '*', '*counter_collection.csv')):
            agg = collections.defaultdict(list)
            for r in csv.DictReader(open(f)):
                k = short(r['Kernel_Name'])
                if 'lp_gemm' in k or 'lp_direct' in k or 'score_fwd' in k or 'lp_split' in k or 'recheck' in k or 'fsub' in k or 'query_pipeline' in k:
                    agg[(r['Counter_Name'], k)].append(float(r['Counter_Value']))
            for (c, k), v in sorted(agg.items()):
                print('| %s | %s | %d | %.6g |' % (c, k, len(v), sum(v) / len(v)))


if __name__ == '__main__':
    main(sys.argv[1])
