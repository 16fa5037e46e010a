#!/usr/bin/env python
"""Copy the judged artefacts of a tools/profile_round.sh run from gpurun_out/ into profiles/<tag>/
(bench JSON lines, SUMMARY.md, rocprofv3 kernel stats of every traced workload, compact per-kernel PMC table)
and refresh profiles/traffic.json:  python tools/collect_profiles.py r04"""
import collections
import csv
import glob
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main(tag):
    src = os.path.join(ROOT, 'gpurun_out', 'profiles_' + tag)
    dst = os.path.join(ROOT, 'profiles', tag)
    os.makedirs(dst, exist_ok=True)
    for f in glob.glob(os.path.join(src, 'bench_*.json')):
        if os.path.getsize(f) > 0:
            shutil.copy(f, dst)
    shutil.copy(os.path.join(src, 'SUMMARY.md'), dst)
    # kernel-trace --stats of the bench command ('trace') and of every evaluate-only trace ('trace_<workload>')
    for d in sorted(glob.glob(os.path.join(src, 'trace*'))):
        if not os.path.isdir(d):
            continue
        stats = glob.glob(os.path.join(d, '*kernel_stats.csv')) + glob.glob(os.path.join(d, '*', '*kernel_stats.csv'))
        if not stats:
            continue
        name = os.path.basename(d)
        out = 'kernel_stats_bench_transe_fb15k237.csv' if name == 'trace' else \
            'kernel_stats_evaluate_only_%s.csv' % ({'trace_eval': 'transe_fb15k237',
                                                    'trace_eval_three_products': 'transe_fb15k237_three_products',
                                                    'trace_l1': 'transe_l1_fb15k237', 'trace_l2direct': 'transe_fb15k237_l2direct',
                                                    'trace_transd': 'transd_fb15k237', 'trace_transh': 'transh_fb15k237'}
                                                   .get(name, name[len('trace_'):]))
        shutil.copy(stats[0], os.path.join(dst, out))
    for f in glob.glob(os.path.join(src, 'timeline_*.txt')):      # (r05) one evaluate() dispatch by dispatch, per workload
        shutil.copy(f, dst)
    for extra in ('level1_kernel_ab.txt', 'hi_stream_probes.txt', 'power_probe_hi_stream.txt', 'n2_dryrun_gloo.txt'):
        if os.path.exists(os.path.join(src, extra)) and os.path.getsize(os.path.join(src, extra)) > 0:
            shutil.copy(os.path.join(src, extra), dst)
    for extra in ('topk_inference.jsonl', 'first_call.json'):
        if os.path.exists(os.path.join(src, extra)) and os.path.getsize(os.path.join(src, extra)) > 0:
            shutil.copy(os.path.join(src, extra), dst)
    if os.path.exists(os.path.join(src, 'power_probe.txt')):
        shutil.copy(os.path.join(src, 'power_probe.txt'), os.path.join(dst, 'power_probe_round_end.txt'))
    rows = []
    for d in sorted(glob.glob(os.path.join(src, 'pmc*'))):
        if not os.path.isdir(d):
            continue
        run = os.path.basename(d)
        for f in glob.glob(os.path.join(d, '*counter_collection.csv')) + glob.glob(os.path.join(d, '*', '*counter_collection.csv')):
            agg = collections.defaultdict(list)
            for r in csv.DictReader(open(f)):
                k = r['Kernel_Name'].replace('(anonymous namespace)::', '').split('(')[0][:60]
                agg[(r['Counter_Name'], k)].append(float(r['Counter_Value']))
            for (c, k), v in sorted(agg.items()):
                rows.append((run, c, k, len(v), sum(v) / len(v)))
    with open(os.path.join(dst, 'pmc_per_kernel_bench_transe_fb15k237.csv'), 'w') as o:
        w = csv.writer(o)
        w.writerow(['run', 'counter', 'kernel', 'launches', 'mean_per_launch'])
        for r in rows:
            w.writerow(r)

    def mean(run_prefix, counter, kern):
        """per-launch mean, SUMMED over the instantiations of a kernel (the count kernel runs as two per evaluate:
        single-query and grouped columns)"""
        got = [m for run, c, k, n, m in rows if run.startswith(run_prefix) and c == counter and kern in k]
        return sum(got) if got else None
    # profiles/traffic.json: the static fallback of bench.py's `roofline.traffic` (used only when the run's own
    # rocprofv3 passes fail); refreshed from THIS round's counters, with the source file named
    tfile = os.path.join(ROOT, 'profiles', 'traffic.json')
    t = json.load(open(tfile)) if os.path.exists(tfile) else {}
    src_csv = 'profiles/%s/pmc_per_kernel_bench_transe_fb15k237.csv' % tag
    for key, run_prefix, kern, what in (
            ('transe_fb15k237', 'pmc_level1', 'lp_split_count', 'lp_split_count_kernel on the ONE-PRODUCT level (run pmc_level1_*: '
             '--split-level 1, what a fitted model is evaluated on)'),
            ('transe_fb15k237:three-products', 'pmc_level0', 'lp_split_count', 'lp_split_count_kernel, three-product sweep (run pmc_level0_*)'),
            ('transe_fb15k237:no-split', 'pmc_nosplit', 'lp_gemm_kernel', 'lp_gemm_kernel<count, L2_EXPAND> (run pmc_nosplit_*: --no-split)')):
        f, wr = mean(run_prefix, 'FETCH_SIZE', kern), mean(run_prefix, 'WRITE_SIZE', kern)
        if f is not None and wr is not None:
            # FETCH_SIZE / WRITE_SIZE are in KB; FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 note)
            t[key] = {'bytes_per_launch': int((2 * f + wr) * 1024), 'fetch_size_kb_raw': f, 'write_size_kb_raw': wr,
                      'note': 'rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes), mean per launch summed over the '
                              'instantiations of %s; one launch = both sides of the batch (2 x 20466 queries x 14541 '
                              'candidates); FETCH_SIZE doubled per MI355X_MICROARCH.md gfx950 note; L2 misses served mostly by '
                              'the 256 MiB Infinity Cache; source %s' % (what, src_csv)}
    json.dump(t, open(tfile, 'w'), indent=1)
    for f in sorted(glob.glob(os.path.join(dst, 'bench_*.json'))):
        try:
            j = json.loads(open(f).read().strip().splitlines()[-1])
        except Exception:
            print('%-44s unreadable' % os.path.basename(f))
            continue
        r = j.get('roofline') or {}
        print('%-52s %8.4f ms  %.3g %s  %s %.1f (%.3f)' % (os.path.basename(f), j['ms_per_step'], j['value'], j['unit'],
                                                           (r.get('kernel') or '')[:24], r.get('achieved', 0), r.get('frac', 0)))


if __name__ == '__main__':
    main(sys.argv[1] if len(sys.argv) > 1 else 'r04')
