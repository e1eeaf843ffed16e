#!/usr/bin/env python3
"""
Turn a tools/profile.sh output directory (gpurun_out/prof_<tag>) into the
committed evidence under profiles/:

  <tag>_summary.md          kernel-trace statistics of the driver's command,
                            the metric kernel's launches INSIDE the timed
                            region picked out of the per-dispatch rows, and
                            the per-launch PMC averages of every workload
  <tag>_kernel_stats.csv    rocprofv3's own --stats table (top rows)
  traffic_<workload>.json   what bench.py reports as roofline.traffic

    python tools/summarize_profile.py r02 config3 headline config4 config5
"""
import collections
import csv
import glob
import json
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)

from pyremap_amd.engine import ABI_VERSION  # noqa: E402


def newest(paths):
    """gpurun MERGES a run's files into gpurun_out/: files of earlier runs
    (other PIDs in their names) may still lie beside the current ones.  Keep
    the newest file of every directory."""
    best = {}
    for q in paths:
        d = os.path.dirname(q)
        if d not in best or os.path.getmtime(q) > os.path.getmtime(best[d]):
            best[d] = q
    return sorted(best.values())


def short(name):
    name = name.replace('(anonymous namespace)::', '').replace('void ', '')
    return name.split('(')[0][:70]


def main():
    tag, workloads = sys.argv[1], sys.argv[2:] or ['config3']
    src = os.path.join(REPO, 'gpurun_out', f'prof_{tag}')
    out = os.path.join(REPO, 'profiles')
    os.makedirs(out, exist_ok=True)
    L = [f'# rocprofv3 summary {tag}', '',
         'Collected by tools/profile.sh.  Pass 1: `rocprofv3 --kernel-trace '
         '--stats -- python3 bench.py --gpus 1 --steps 20 --warmup 5` (the '
         "driver's command).  Pass 2..n: one `rocprofv3 --pmc <group>` pass "
         'per counter group and workload over `bench.py --workload W '
         '--no-extra --no-cpu`.', '']

    # ---- pass 1: the driver's command ---------------------------------
    live = None
    try:
        live = json.loads(open(os.path.join(
            src, 'trace_bench.json')).read().strip().splitlines()[-1])
    except (OSError, ValueError, IndexError):
        pass
    stats = newest(glob.glob(os.path.join(src, 'trace', '*',
                                          '*_kernel_stats.csv')))
    if stats:
        rows = list(csv.DictReader(open(stats[0])))
        with open(os.path.join(out, f'{tag}_kernel_stats.csv'), 'w') as f:
            w = csv.DictWriter(f, fieldnames=rows[0].keys())
            w.writeheader()
            for r in rows[:14]:
                w.writerow(r)
        L += ['## rocprofv3 --kernel-trace --stats (remap kernels, whole run)',
              '', '| kernel | calls | avg ns | min ns | max ns |',
              '|---|---|---|---|---|']
        for r in [r for r in rows if 'remap::' in r['Name']][:8]:
            L.append(f"| `{short(r['Name'])}` | {r['Calls']} | "
                     f"{float(r['AverageNs']):.0f} | {r['MinNs']} | "
                     f"{r['MaxNs']} |")
        L.append('')
    trace = newest(glob.glob(os.path.join(src, 'trace', '*',
                                          '*_kernel_trace.csv')))
    timed_avg = None
    if trace and live:
        steps, warmup = live['steps'], live['warmup']
        rows = [r for r in csv.DictReader(open(trace[0]))
                if 'spmm_' in r['Kernel_Name']]
        rows.sort(key=lambda r: int(r['Start_Timestamp']))
        # The metric workload's launches are the LAST run of at least W
        # warm-up + K timed + min(K, 50) individually timed + 100 steady
        # consecutive launches of one kernel on one grid (what follows --
        # graph replays of the short extras, the host-buffer extra -- uses
        # other kernels or shorter runs).
        tail = 100 + min(steps, 50)
        block = tail + steps + warmup
        runs, start = [], 0
        for n in range(1, len(rows) + 1):
            if n == len(rows) or \
                    (rows[n]['Kernel_Name'], rows[n]['Grid_Size_X']) != \
                    (rows[start]['Kernel_Name'], rows[start]['Grid_Size_X']):
                runs.append((start, n))
                start = n
        # (the graph replays of the short extras are long runs too: keep
        # to the kernel family the bench line names)
        family = (live['roofline'].get('kernel') or 'spmm_').split()[0]
        long_runs = [r for r in runs if r[1] - r[0] >= block and
                     family in rows[r[0]]['Kernel_Name']]
        first, last = long_runs[-1] if long_runs else runs[-1]
        mine = rows[first:last]
        main_grid = mine[-1]['Grid_Size_X']
        dur = [int(r['End_Timestamp']) - int(r['Start_Timestamp'])
               for r in mine]
        cand = dur[-block:] if len(dur) >= block else dur
        timed = cand[warmup:warmup + steps]
        timed_avg = sum(timed) / len(timed)
        L += ['## The metric kernel inside the timed region (pass 1, '
              'per-dispatch rows)', '',
              f'Kernel `{short(mine[-1]["Kernel_Name"])}`, grid '
              f'{main_grid} x {mine[-1]["Workgroup_Size_X"]}, '
              f'rocprofv3 columns VGPR_Count {mine[-1]["VGPR_Count"]} / '
              f'SGPR_Count {mine[-1]["SGPR_Count"]} (the code object\'s '
              f'.vgpr_count, which sets the occupancy, is what '
              f'tools/kernel_regs.py prints): the {steps} launches of the timed region average '
              f'**{timed_avg:.0f} ns** (min {min(timed)}, max {max(timed)}); '
              f'the bench line of the same run reports kernel_ms_mean = '
              f'{live["roofline"]["kernel_ms_mean"] * 1e6:.0f} ns (one HIP '
              f'event pair around the region, i.e. including the gaps '
              f'between launches), frac {live["roofline"]["frac"]:.4f}.  '
              f'**rocprof per-dispatch mean / bench kernel_ms_mean of the '
              f'same run = '
              f'{timed_avg / (live["roofline"]["kernel_ms_mean"] * 1e6):.4f}'
              f'**: multiply another box\'s `ms_per_step` / kernel_ms_mean '
              f'by it for that box\'s per-dispatch figure (boxes differ by '
              f'up to 9 % on this launch; the ratio does not).', '']
        with open(os.path.join(out, f'{tag}_bench_under_rocprof.json'),
                  'w') as f:
            json.dump(live, f, indent=1)

    # ---- PMC passes -----------------------------------------------------
    for wl in workloads:
        pmc = collections.OrderedDict()
        cfg = None
        for path in newest(glob.glob(os.path.join(
                src, f'pmc_{wl}_*', '*', '*_counter_collection.csv'))):
            per = collections.defaultdict(list)
            names = collections.Counter()
            for r in csv.DictReader(open(path)):
                if 'spmm_' in r['Kernel_Name']:
                    per[r['Counter_Name']].append(float(r['Counter_Value']))
                    names[short(r['Kernel_Name'])] += 1
            for name, vals in per.items():
                vals = vals[3:] if len(vals) > 6 else vals   # skip warm-up
                pmc[name] = (sum(vals) / len(vals), min(vals), max(vals),
                             len(vals), names.most_common(1)[0][0])
        for path in glob.glob(os.path.join(src, f'pmc_{wl}_*.json')):
            try:
                cfg = json.loads(open(path).read().strip().splitlines()[-1])
                break
            except (OSError, ValueError, IndexError):
                continue
        if not pmc:
            continue
        L += [f'## rocprofv3 --pmc, {wl} (per launch of the remap kernel)',
              '', '| counter | mean | min | max | launches |',
              '|---|---|---|---|---|']
        for name, (mean, lo, hi, n, _) in pmc.items():
            L.append(f'| {name} | {mean:.6g} | {lo:.6g} | {hi:.6g} | {n} |')
        L.append('')
        if 'FETCH_SIZE' in pmc and 'WRITE_SIZE' in pmc:
            # FETCH_SIZE / WRITE_SIZE are in KiB.  MI355X_MICROARCH.md (HBM):
            # on gfx950 FETCH_SIZE reports exactly 1/2 of the bytes of a wide
            # (16 B/lane) coalesced streaming read -> double it; WRITE_SIZE
            # is exact for 16 B/lane streaming stores.
            fetch = pmc['FETCH_SIZE'][0] * 1024 * 2
            write = pmc['WRITE_SIZE'][0] * 1024
            K = cfg['config']['fields_K'] if cfg else None
            mode = cfg['config']['mode'] if cfg else None
            alg = cfg['roofline']['bytes_alg_per_launch'] if cfg else None
            traffic = dict(
                workload=wl, K=K, mode=mode,
                fetch_bytes_per_launch=fetch, write_bytes_per_launch=write,
                hbm_bytes_per_launch=fetch + write,
                bytes_alg_per_launch=alg,
                traffic_over_algorithmic=(fetch + write) / alg if alg else
                None,
                fetch_size_raw_kib=pmc['FETCH_SIZE'][0],
                write_size_raw_kib=pmc['WRITE_SIZE'][0],
                kernel=pmc['FETCH_SIZE'][4],
                abi_version=ABI_VERSION,
                kernel_ms_mean_under_pmc=cfg['roofline']['kernel_ms_mean']
                if cfg else None,
                source=f'profiles/{tag}_summary.md: rocprofv3 --pmc '
                       f'FETCH_SIZE and --pmc WRITE_SIZE (separate passes); '
                       f'FETCH_SIZE x 1024 x 2 (gfx950 half-count correction '
                       f'for 16 B/lane reads, MI355X_MICROARCH.md) + '
                       f'WRITE_SIZE x 1024')
            # the raster-numbered files of round 2 keep their names; other
            # numberings of the synthetic source mesh carry theirs
            loc = cfg['config'].get('locality', 'raster') if cfg else 'raster'
            suffix = '' if loc == 'raster' else f'_{loc}'
            if mode not in (None, 'fracb'):
                suffix += f'_{mode}'
            traffic['numbering'] = loc
            with open(os.path.join(out, f'traffic_{wl}{suffix}.json'),
                      'w') as f:
                json.dump(traffic, f, indent=1)
            L.append(f'Corrected fabric traffic per launch: reads '
                     f'{fetch / 1e9:.3f} GB (FETCH_SIZE x 2), writes '
                     f'{write / 1e9:.3f} GB, total '
                     f'{(fetch + write) / 1e9:.3f} GB' +
                     (f' = {(fetch + write) / alg:.3f} x the algorithmic '
                      f'{alg / 1e9:.3f} GB' if alg else '') + '.')
            if 'TCC_HIT_sum' in pmc:
                hit, miss = pmc['TCC_HIT_sum'][0], pmc['TCC_MISS_sum'][0]
                L.append(f'L2 hit rate (all requests): '
                         f'{hit / (hit + miss):.3f}.')
            L.append('')
    with open(os.path.join(out, f'{tag}_summary.md'), 'w') as f:
        f.write('\n'.join(L) + '\n')
    print('\n'.join(L))


if __name__ == '__main__':
    main()
