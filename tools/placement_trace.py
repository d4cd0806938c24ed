"""Timeline of ONE optimal_placement from a rocprofv3 --kernel-trace CSV (round 5): span from its first to its last kernel, the
time inside kernels by name, and the idle time BEHIND each kind of kernel (host round trips, launch latency).
usage: placement_trace.py <kernel_trace.csv>   -- the last complete placement of the run is taken (they start with
qr_tops_from_norms_kernel or a norms kernel and contain qr_* kernels only)."""
import csv
import sys
from collections import defaultdict

rows = []
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name']))
rows.sort()


def short(n):
    n = n.replace('(anonymous namespace)::', '').replace('void ', '')
    return n.split('(')[0][:48]


# placements: from the kernel that opens one (qr_tops_from_norms_kernel, or the initialising sweep qr_refresh_*<.., true>) to the
# last qr_* kernel before the next
starts = [i for i, (s, e, n) in enumerate(rows)
          if 'qr_tops_from_norms' in n or ('qr_refresh' in n and ', true>(' in n)]
runs = []
for a, b in zip(starts, starts[1:] + [len(rows)]):
    seg = [(s, e, short(n)) for s, e, n in rows[a:b]]
    while seg and 'qr_' not in seg[-1][2]:
        seg.pop()
    if len(seg) > 20:
        runs.append(seg)
print(f'{len(runs)} placements in the trace; the last three:')
for run in runs[-3:]:
    span = (run[-1][1] - run[0][0]) / 1e3
    busy = defaultdict(float)
    gap_after = defaultdict(float)
    cnt = defaultdict(int)
    for i, (s, e, n) in enumerate(run):
        busy[n] += (e - s) / 1e3
        cnt[n] += 1
        if i + 1 < len(run):
            gap_after[n] += max(0, run[i + 1][0] - e) / 1e3
    tb, tg = sum(busy.values()), sum(gap_after.values())
    print(f'  span {span / 1e3:.3f} ms = kernels {tb / 1e3:.3f} + idle {tg / 1e3:.3f}')
    for n in sorted(busy, key=lambda k: -busy[k] - gap_after[k]):
        print(f'    {n:50s} x{cnt[n]:3d}  busy {busy[n] / 1e3:7.3f} ms   idle behind it {gap_after[n] / 1e3:7.3f} ms')
