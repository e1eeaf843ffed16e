#!/usr/bin/env python3
"""
Group rocprofv3 --pmc counter rows of the spmm kernels by sweep variant.

    python tools/pmc_report.py <log with PMCVARIANT lines> <counter csv>...

tools/sweep.py --pmc N launches every variant N times in order; the CSV has
one row per dispatch and counter, so dispatch j of the spmm kernels belongs to
variant j // N.
"""
import collections
import csv
import sys


def main():
    log, csvs = sys.argv[1], sys.argv[2:]
    names = [line.split(' ', 2)[2].strip() for line in open(log)
             if line.startswith('PMCVARIANT')]
    table = collections.defaultdict(dict)
    for path in csvs:
        rows = [r for r in csv.DictReader(open(path))
                if 'spmm_' in r['Kernel_Name']]
        per_counter = collections.defaultdict(list)
        for r in rows:
            per_counter[r['Counter_Name']].append(
                (int(r['Dispatch_Id']), float(r['Counter_Value'])))
        for cname, vals in per_counter.items():
            vals.sort()
            n = len(vals) // max(len(names), 1)
            for vi, name in enumerate(names):
                chunk = [v for _, v in vals[vi * n:(vi + 1) * n]]
                chunk = chunk[1:] if len(chunk) > 1 else chunk
                table[name][cname] = sum(chunk) / len(chunk)
    counters = sorted({c for t in table.values() for c in t})
    print('variant'.ljust(52) + ''.join(c[:18].rjust(20) for c in counters))
    for name in names:
        print(name[:50].ljust(52) + ''.join(
            f'{table[name].get(c, float("nan")):20.4g}' for c in counters))


if __name__ == '__main__':
    main()
