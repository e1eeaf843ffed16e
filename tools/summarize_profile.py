#!/usr/bin/env python3
"""
Turn a tools/profile.sh output directory (gpurun_out/prof_<tag>) into the
committed evidence under profiles/: the rocprofv3 kernel-stats rows, the
per-launch PMC averages of the remap kernel, and profiles/traffic_<workload>
.json which bench.py reports as roofline.traffic.

    python tools/summarize_profile.py r01 config3 512 fracb
"""
import collections
import csv
import glob
import json
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    tag, workload, K, mode = sys.argv[1], sys.argv[2], int(sys.argv[3]), \
        sys.argv[4]
    src = os.path.join(REPO, 'gpurun_out', f'prof_{tag}')
    out = os.path.join(REPO, 'profiles')
    os.makedirs(out, exist_ok=True)
    lines = [f'# rocprofv3 summary {tag} ({workload}, K = {K}, mode {mode})',
             '',
             'Collected by tools/profile.sh: the kernel-trace pass runs '
             '`bench.py --no-cpu --no-extra` (default steps / warm-up, so its '
             'average is comparable with `roofline.kernel_ms_mean` of the '
             'bench line taken on the same box); each PMC group is its own '
             '`rocprofv3 --pmc ...` pass over `bench.py --steps 20 --warmup '
             '3 --no-cpu --no-extra`.', '']

    # gpurun MERGES a run's files into gpurun_out/: files of earlier runs
    # under the same tag may still be there.  tools/profile.sh ends with a
    # listing of what THIS run wrote -- keep to it when it is available.
    listing = os.path.join(REPO, 'gpurun_out', f'profile_{tag}.log')
    current = None
    if os.path.exists(listing):
        current = {os.path.basename(line.strip())
                   for line in open(listing) if line.strip().endswith('.csv')}

    def only_current(paths):
        if not current:
            return paths
        return [q for q in paths if os.path.basename(q) in current]

    stats = only_current(glob.glob(
        os.path.join(src, 'trace', '*', '*_kernel_stats.csv')))
    kernel_avg_ns = None
    kernel_name = None
    if stats:
        rows = list(csv.DictReader(open(stats[0])))
        keep = [r for r in rows if 'remap::' in r['Name']][:6]
        with open(os.path.join(out, f'{tag}_kernel_stats.csv'), 'w') as f:
            w = csv.DictWriter(f, fieldnames=rows[0].keys())
            w.writeheader()
            for r in rows[:12]:
                w.writerow(r)
        lines.append('## rocprofv3 --kernel-trace --stats (remap kernels)')
        lines.append('')
        lines.append('| kernel | calls | avg ns | min ns | max ns |')
        lines.append('|---|---|---|---|---|')
        for r in keep:
            name = r['Name'].replace('(anonymous namespace)::', '')
            name = name.replace('void ', '')[:60]
            lines.append(f"| `{name}` | {r['Calls']} | "
                         f"{float(r['AverageNs']):.0f} | {r['MinNs']} | "
                         f"{r['MaxNs']} |")
            if 'spmm' in r['Name'] and kernel_avg_ns is None:
                kernel_avg_ns = float(r['AverageNs'])
                kernel_name = name.split('<')[0].replace('remap::', '')
        lines.append('')

    pmc = collections.OrderedDict()
    for path in only_current(sorted(glob.glob(os.path.join(
            src, 'pmc_*', '*', '*_counter_collection.csv')))):
        per = collections.defaultdict(list)
        for r in csv.DictReader(open(path)):
            if 'spmm_' in r['Kernel_Name']:
                per[r['Counter_Name']].append(float(r['Counter_Value']))
        for name, vals in per.items():
            vals = vals[3:] if len(vals) > 6 else vals      # skip warm-up
            pmc[name] = (sum(vals) / len(vals), min(vals), max(vals),
                         len(vals))
    lines.append('## rocprofv3 --pmc (one pass per group), per launch of '
                 'the remap kernel')
    lines.append('')
    lines.append('| counter | mean | min | max | launches |')
    lines.append('|---|---|---|---|---|')
    for name, (mean, lo, hi, n) in pmc.items():
        lines.append(f'| {name} | {mean:.6g} | {lo:.6g} | {hi:.6g} | {n} |')
    lines.append('')

    if 'FETCH_SIZE' in pmc and 'WRITE_SIZE' in pmc:
        # FETCH_SIZE / WRITE_SIZE are in KiB.  MI355X_MICROARCH.md (HBM):
        # on gfx950 FETCH_SIZE reports exactly 1/2 of the bytes of a wide
        # (16 B/lane) coalesced streaming read -> double it; WRITE_SIZE is
        # exact for 16 B/lane streaming stores.
        fetch = pmc['FETCH_SIZE'][0] * 1024 * 2
        write = pmc['WRITE_SIZE'][0] * 1024
        traffic = dict(
            workload=workload, K=K, mode=mode,
            fetch_bytes_per_launch=fetch, write_bytes_per_launch=write,
            hbm_bytes_per_launch=fetch + write,
            fetch_size_raw_kib=pmc['FETCH_SIZE'][0],
            write_size_raw_kib=pmc['WRITE_SIZE'][0],
            kernel_avg_ns_rocprof=kernel_avg_ns, kernel=kernel_name,
            source=f'profiles/{tag}_summary.md: rocprofv3 --pmc FETCH_SIZE '
                   f'and --pmc WRITE_SIZE (separate passes); FETCH_SIZE x '
                   f'1024 x 2 (gfx950 half-count correction for 16 B/lane '
                   f'reads, MI355X_MICROARCH.md) + WRITE_SIZE x 1024')
        with open(os.path.join(out, f'traffic_{workload}.json'), 'w') as f:
            json.dump(traffic, f, indent=1)
        lines.append(f'Corrected fabric traffic per launch: reads '
                     f'{fetch / 1e9:.3f} GB (FETCH_SIZE x 2), writes '
                     f'{write / 1e9:.3f} GB, total '
                     f'{(fetch + write) / 1e9:.3f} GB.')
        if 'TCC_HIT_sum' in pmc:
            hit, miss = pmc['TCC_HIT_sum'][0], pmc['TCC_MISS_sum'][0]
            lines.append(f'L2 hit rate (all requests): '
                         f'{hit / (hit + miss):.3f}.')
    with open(os.path.join(out, f'{tag}_summary.md'), 'w') as f:
        f.write('\n'.join(lines) + '\n')
    print('\n'.join(lines))


if __name__ == '__main__':
    main()
