#!/usr/bin/env python
"""Copy the judged artefacts of a tools/profile_round.sh run from gpurun_out/ into profiles/<tag>/
(bench JSON lines, SUMMARY.md, rocprofv3 kernel stats, compact per-kernel PMC table) and refresh
profiles/traffic.json:  python tools/collect_profiles.py r01"""
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
        shutil.copy(f, dst)
    shutil.copy(os.path.join(src, 'SUMMARY.md'), dst)
    shutil.copy(os.path.join(src, 'trace', 'bench_kernel_stats.csv'),
                os.path.join(dst, 'kernel_stats_bench_transe_fb15k237.csv'))
    if os.path.exists(os.path.join(src, 'trace_eval', 'bench_kernel_stats.csv')):
        shutil.copy(os.path.join(src, 'trace_eval', 'bench_kernel_stats.csv'),
                    os.path.join(dst, 'kernel_stats_evaluate_only_transe_fb15k237.csv'))
    if os.path.exists(os.path.join(src, 'trace_l1', 'bench_kernel_stats.csv')):
        shutil.copy(os.path.join(src, 'trace_l1', 'bench_kernel_stats.csv'),
                    os.path.join(dst, 'kernel_stats_evaluate_only_transe_l1_fb15k237.csv'))
    for extra in ('topk_inference.jsonl',):
        if os.path.exists(os.path.join(src, extra)):
            shutil.copy(os.path.join(src, extra), dst)
    if os.path.exists(os.path.join(src, 'power_probe.txt')):
        shutil.copy(os.path.join(src, 'power_probe.txt'), os.path.join(dst, 'power_probe_round_end.txt'))
    rows = []
    for d in sorted(glob.glob(os.path.join(src, 'pmc*'))):
        if not os.path.isdir(d):
            continue
        for f in glob.glob(os.path.join(d, '*counter_collection.csv')) + glob.glob(os.path.join(d, '*', '*counter_collection.csv')):
            agg = collections.defaultdict(list)
            for r in csv.DictReader(open(f)):
                k = r['Kernel_Name'].replace('(anonymous namespace)::', '').split('(')[0][:60]
                agg[(r['Counter_Name'], k)].append(float(r['Counter_Value']))
            for (c, k), v in sorted(agg.items()):
                rows.append((c, k, len(v), sum(v) / len(v)))
    with open(os.path.join(dst, 'pmc_per_kernel_bench_transe_fb15k237.csv'), 'w') as o:
        w = csv.writer(o)
        w.writerow(['counter', 'kernel', 'launches', 'mean_per_launch'])
        for r in rows:
            w.writerow(r)

    def mean(counter, kern):
        """per-launch mean, SUMMED over the instantiations of a kernel (the count kernel runs as two per evaluate:
        single-query and grouped columns)"""
        got = [m for c, k, n, m in rows if c == counter and kern in k]
        return sum(got) if got else None
    tfile = os.path.join(ROOT, 'profiles', 'traffic.json')
    t = json.load(open(tfile))
    for key, kern in (('transe_fb15k237', 'lp_split_count'), ('transe_fb15k237:no-split', 'lp_gemm_kernel')):
        f, wr = mean('FETCH_SIZE', kern), mean('WRITE_SIZE', kern)
        if f is not None and wr is not None and key in t:
            # FETCH_SIZE / WRITE_SIZE are in KB; FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 note)
            t[key].update({'bytes_per_launch': int((2 * f + wr) * 1024), 'fetch_size_kb_raw': f, 'write_size_kb_raw': wr})
    json.dump(t, open(tfile, 'w'), indent=1)
    for f in sorted(glob.glob(os.path.join(dst, 'bench_*.json'))):
        j = json.loads(open(f).read().strip().splitlines()[-1])
        print('%-44s %8.4f ms  %.3g %s  %s %.1f (%.3f)' % (os.path.basename(f), j['ms_per_step'], j['value'], j['unit'],
                                                           j['roofline']['kernel'][:24], j['roofline']['achieved'],
                                                           j['roofline']['frac']))


if __name__ == '__main__':
    main(sys.argv[1] if len(sys.argv) > 1 else 'r02')
