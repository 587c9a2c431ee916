#!/usr/bin/env python3
"""Fold the per-case rocprofv3 runs of tools/profile_aux.sh into one JSON table.

    python tools/aux_report.py gpurun_out/r02_aux

Per bench_aux case: the library's own kernels (everything that is not an ATen / runtime-copy kernel) with their mean
duration, the kernel-only time of one call (sum over the kernels one call launches), algorithmic GB/s on that time, and the
HBM bytes of one call from the PMC counters: FETCH_SIZE (KB; doubled, gfx950 reports half the bytes of wide coalesced
reads -- MI355X_MICROARCH.md, HBM section) and WRITE_SIZE (KB)."""
import csv
import glob
import json
import os
import re
import sys

CALLS = 23                      # bench_aux.timeit: 3 warm-up + 20 timed calls per case


def short(name):
    name = re.sub(r'\(anonymous namespace\)::', '', name)
    name = re.sub(r'^void ', '', name)
    m = re.match(r'([A-Za-z0-9_:]+(?:<[^(]*>)?)\(', name)
    return m.group(1) if m else name[:80]


def ours(name):
    return not (name.startswith('at::') or 'rocclr' in name or name.startswith('Cijk') or 'elementwise' in name)


def kernel_times(d):
    out = {}
    for path in glob.glob(os.path.join(d, '**', '*_kernel_trace.csv'), recursive=True):
        for row in csv.DictReader(open(path)):
            k = short(row['Kernel_Name'])
            if not ours(k):
                continue
            e = out.setdefault(k, {'calls': 0, 'total_us': 0.0, 'workgroups': int(row['Grid_Size_X']) // max(int(row['Workgroup_Size_X']), 1),
                                   'threads': int(row['Workgroup_Size_X'])})
            e['calls'] += 1
            e['total_us'] += (int(row['End_Timestamp']) - int(row['Start_Timestamp'])) / 1e3
    return out


def counter_sum(d, counter):
    tot = 0.0
    for path in glob.glob(os.path.join(d, '**', '*_counter_collection.csv'), recursive=True):
        for row in csv.DictReader(open(path)):
            if row['Counter_Name'] == counter and ours(short(row['Kernel_Name'])):
                tot += float(row['Counter_Value'])
    return tot


def main():
    base = sys.argv[1]
    wall = {}
    for ln in open(base + '_wall.jsonl'):
        if ln.startswith('{'):
            e = json.loads(ln)
            wall[e['case']] = e
    table = {}
    for case, w in wall.items():
        kt = kernel_times(os.path.join(base + '_kt', case))
        per_call_us = sum(e['total_us'] for e in kt.values()) / CALLS
        fetch = counter_sum(os.path.join(base + '_fetch', case), 'FETCH_SIZE') / CALLS * 1024 * 2
        write = counter_sum(os.path.join(base + '_write', case), 'WRITE_SIZE') / CALLS * 1024
        nbytes = w['algorithmic_bytes']
        table[case] = {
            'kernels': {k: {'launches_per_call': round(e['calls'] / CALLS, 2), 'mean_us': round(e['total_us'] / e['calls'], 2),
                            'workgroups': e['workgroups'], 'threads': e['threads']} for k, e in kt.items()},
            'kernel_us_per_call': round(per_call_us, 2),
            'wall_us_per_call': round(w['ms'] * 1e3, 2),
            'algorithmic_bytes': nbytes, 'bytes_counted': w['bytes_counted'],
            'kernel_GBps': round(nbytes / per_call_us / 1e3, 1) if per_call_us else None,
            'frac_of_8TBps_kernel': round(nbytes / per_call_us / 1e3 / 8000.0, 3) if per_call_us else None,
            'frac_of_8TBps_wall': w['frac_of_8TB/s'],
            'hbm_fetch_bytes_per_call': round(fetch), 'hbm_write_bytes_per_call': round(write),
            'traffic_over_algorithmic': round((fetch + write) / nbytes, 2) if nbytes else None,
        }
    out = {'source': 'tools/profile_aux.sh: rocprofv3 --kernel-trace / --pmc FETCH_SIZE / --pmc WRITE_SIZE, one run per case',
           'notes': 'FETCH_SIZE doubled (gfx950 counts half of wide coalesced reads); Infinity-Cache hits count as fetches; '
                    'kernel_us_per_call sums every library kernel one call launches',
           'cases': table}
    json.dump(out, open(base + '_kernels.json', 'w'), indent=1)
    for c, e in table.items():
        print(f"{c:44s} kernel {e['kernel_us_per_call']:8.1f} us  {e['kernel_GBps'] or 0:7.0f} GB/s  frac {e['frac_of_8TBps_kernel']}  traffic x{e['traffic_over_algorithmic']}")


if __name__ == '__main__':
    main()
