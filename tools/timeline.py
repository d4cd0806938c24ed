#!/usr/bin/env python3
"""Device timeline of ONE steady-state bench step from rocprofv3 traces (CPU tool).

    rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d <dir> -o run -- python3 bench.py ...
    python3 tools/timeline.py <dir> [--anchor stats_gram] [--step -2]

Reads every *_kernel_trace.csv / *_memory_copy_trace.csv under <dir>, orders the records by start time and prints the
records between two consecutive launches of the anchor kernel (default: the Gram pass; --step picks which interval, -2 =
the last complete one): start offset, duration and the idle gap in front of every record, then a summary -- busy time per
kernel name, total idle, the number of fill / copy records.  This is how the time OUTSIDE the three big kernels of a step
is itemised."""
import argparse
import csv
import glob
import os
import re
from collections import defaultdict


def short(name):
    name = re.sub(r'\(anonymous namespace\)::', '', name)
    name = re.sub(r'^void ', '', name)
    name = re.sub(r'\(.*', '', name)
    return name[:70]


def load(d):
    recs = []
    for path in glob.glob(os.path.join(d, '**', '*_trace.csv'), recursive=True):
        with open(path) as f:
            for row in csv.DictReader(f):
                if 'Start_Timestamp' not in row or 'End_Timestamp' not in row:
                    continue
                label = row.get('Kernel_Name') or ('copy ' + row.get('Direction', row.get('Kind', '?')))
                recs.append((int(row['Start_Timestamp']), int(row['End_Timestamp']), short(label)))
    recs.sort()
    return recs


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('dir')
    ap.add_argument('--anchor', default='stats_gram')
    ap.add_argument('--step', type=int, default=-2)
    ap.add_argument('--min-us', type=float, default=0.0, help='fold records shorter than this into one line per name')
    a = ap.parse_args()
    recs = load(a.dir)
    idx = [i for i, r in enumerate(recs) if a.anchor in r[2] and 'rowmean_stats' not in r[2] and 'finalize' not in r[2]]
    if len(idx) < 2:
        raise SystemExit(f'fewer than two {a.anchor} launches in {a.dir} ({len(recs)} records)')
    k = a.step if a.step >= 0 else len(idx) - 1 + a.step
    i0, i1 = idx[k], idx[k + 1]
    t0 = recs[i0][0]
    print(f'step between anchor launches {k} and {k + 1} of {len(idx)}: {(recs[i1][0] - t0) / 1e6:.3f} ms, '
          f'{i1 - i0} records')
    busy = defaultdict(lambda: [0, 0.0])
    idle = 0.0
    prev_end = recs[i0][0]
    print(f'{"start ms":>9} {"dur us":>10} {"gap us":>9}  name')
    for s, e, name in recs[i0:i1]:
        gap = (s - prev_end) / 1e3
        if gap > 0:
            idle += gap
        dur = (e - s) / 1e3
        busy[name][0] += 1
        busy[name][1] += dur
        if dur >= a.min_us or gap > 20:
            print(f'{(s - t0) / 1e6:9.3f} {dur:10.1f} {gap:9.1f}  {name}')
        prev_end = max(prev_end, e)
    tail = (recs[i1][0] - prev_end) / 1e3
    idle += max(tail, 0.0)
    print(f'{"":>9} {"":>10} {tail:9.1f}  (gap before the next anchor)')
    print('-- busy per name')
    for name, (cnt, us) in sorted(busy.items(), key=lambda kv: -kv[1][1]):
        print(f'{us / 1e3:9.3f} ms  x{cnt:<4d} {name}')
    print(f'-- idle {idle / 1e3:.3f} ms of {(recs[i1][0] - t0) / 1e6:.3f} ms')
    tot = defaultdict(int)
    for _, _, name in recs:
        if 'fillBuffer' in name or 'copyBuffer' in name or name.startswith('copy '):
            tot[name] += 1
    print('-- whole run: ' + ', '.join(f'{v} x {k}' for k, v in sorted(tot.items())))


if __name__ == '__main__':
    main()
