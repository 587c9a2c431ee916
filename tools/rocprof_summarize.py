#!/usr/bin/env python3
"""Summarise rocprofv3 CSV output directories into one JSON document.

    python tools/rocprof_summarize.py OUT.json DIR [DIR ...]

For every DIR (as written by `rocprofv3 --kernel-trace [--stats | --pmc ...] --output-format csv -d DIR -- prog`):
  *_kernel_trace.csv        -> per (kernel, grid, workgroup, LDS) group: calls, mean/min/max duration in us
  *_counter_collection.csv  -> per group: mean counter value per dispatch (summed over the XCD instances that
                               rocprofv3 reports as separate rows of one dispatch)
Dispatch groups are keyed by the kernel's short name plus its launch geometry, which identifies the workload case of
tools/bench_aux.py and tools/prof_hilbert.py (every case has its own shape).  `skip_first` dispatches per group are
dropped as warm-up when a group has more than 2 * skip_first calls.
"""
import csv
import glob
import json
import os
import re
import sys
from collections import defaultdict


def short(name):
    name = re.sub(r'\(anonymous namespace\)::', '', name)
    name = re.sub(r'^void ', '', name)
    m = re.match(r'([A-Za-z0-9_:]+(?:<[^(]*>)?)\(', name)
    return m.group(1) if m else name[:80]


def key_of(row):
    return (short(row['Kernel_Name']), int(row['Grid_Size_X']) // max(int(row['Workgroup_Size_X']), 1),
            int(row['Workgroup_Size_X']), int(row.get('LDS_Block_Size', 0) or 0))


def summarise_dir(d, skip_first=1):
    out = {}
    for path in glob.glob(os.path.join(d, '**', '*_kernel_trace.csv'), recursive=True):
        groups = defaultdict(list)
        for row in csv.DictReader(open(path)):
            groups[key_of(row)].append((int(row['End_Timestamp']) - int(row['Start_Timestamp'])) / 1e3)
        for k, v in groups.items():
            w = v[skip_first:] if len(v) > 2 * skip_first else v
            e = out.setdefault('|'.join(map(str, k)), {'kernel': k[0], 'workgroups': k[1], 'threads': k[2], 'lds_bytes': k[3]})
            e.update({'calls': len(v), 'mean_us': round(sum(w) / len(w), 3), 'min_us': round(min(w), 3), 'max_us': round(max(w), 3)})
    for path in glob.glob(os.path.join(d, '**', '*_counter_collection.csv'), recursive=True):
        per_dispatch = defaultdict(lambda: defaultdict(float))
        meta = {}
        for row in csv.DictReader(open(path)):
            did = int(row['Dispatch_Id'])
            per_dispatch[did][row['Counter_Name']] += float(row['Counter_Value'])
            if did not in meta:
                meta[did] = (short(row['Kernel_Name']), int(row['Grid_Size']) // max(int(row['Workgroup_Size']), 1),
                             int(row['Workgroup_Size']), int(row.get('LDS_Block_Size', 0) or 0))
        groups = defaultdict(list)
        for did in sorted(per_dispatch):
            groups[meta[did]].append(per_dispatch[did])
        for k, lst in groups.items():
            w = lst[skip_first:] if len(lst) > 2 * skip_first else lst
            e = out.setdefault('|'.join(map(str, k)), {'kernel': k[0], 'workgroups': k[1], 'threads': k[2], 'lds_bytes': k[3]})
            c = e.setdefault('counters', {})
            for name in w[0]:
                c[name] = round(sum(x[name] for x in w) / len(w), 2)
    return out


def main():
    dst, dirs = sys.argv[1], sys.argv[2:]
    merged = {}
    for d in dirs:
        for k, e in summarise_dir(d).items():
            m = merged.setdefault(k, {})
            cnt = m.get('counters', {})
            cnt.update(e.get('counters', {}))
            m.update({kk: vv for kk, vv in e.items() if kk != 'counters'})
            if cnt:
                m['counters'] = cnt
    json.dump({'source_dirs': dirs, 'groups': merged}, open(dst, 'w'), indent=1, sort_keys=True)
    print(f'{len(merged)} dispatch groups -> {dst}')


if __name__ == '__main__':
    main()
