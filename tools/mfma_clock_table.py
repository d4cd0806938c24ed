#!/usr/bin/env python3
"""From a tools/profile_round.sh output directory: per dominant kernel the MFMA-busy share, the held clock and the issue /
wait split (CPU tool).   usage: mfma_clock_table.py <dir> [...]
  clock     = GRBM_GUI_ACTIVE / 8 XCDs / mean kernel duration (kernel-trace stats of the same directory)
  mfma busy = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x GRBM_GUI_ACTIVE / 8)          (cycles the matrix pipe executes)
  wait / issue-stall / active = SQ_WAIT_ANY, SQ_WAIT_INST_ANY, SQ_ACTIVE_INST_ANY over SQ_WAVE_CYCLES (disjoint, quad-cycles)"""
import csv, glob, os, re, sys
from collections import defaultdict


def short(n):
    n = re.sub(r'\(anonymous namespace\)::', '', n).replace('void ', '')
    return re.sub(r'\(.*', '', n)


for d in sys.argv[1:]:
    dur = {}
    for f in glob.glob(os.path.join(d, 'stats', '**', '*kernel_stats.csv'), recursive=True):
        for row in csv.DictReader(open(f)):
            dur[short(row['Name'])] = float(row['AverageNs']) / 1e6
    ctr = defaultdict(lambda: defaultdict(list))
    for f in glob.glob(os.path.join(d, 'SQ*', '**', '*counter_collection.csv'), recursive=True):
        for row in csv.DictReader(open(f)):
            ctr[short(row['Kernel_Name'])][row['Counter_Name']].append(float(row['Counter_Value']))
    print(f'== {d}')
    print(f'{"kernel":58s} {"ms":>8s} {"GHz":>6s} {"mfma busy":>9s} {"wait":>6s} {"issue stall":>11s} {"active":>6s} {"VALU/MFMA":>9s} {"LDS inst/MFMA":>13s}')
    for k, c in sorted(ctr.items(), key=lambda kv: -dur.get(kv[0], 0)):
        if k not in dur or 'GRBM_GUI_ACTIVE' not in c or dur[k] < 0.05:
            continue
        m = {n: sum(v) / len(v) for n, v in c.items()}
        cyc = m['GRBM_GUI_ACTIVE'] / 8
        n_mfma = m.get('SQ_VALU_MFMA_BUSY_CYCLES', 0) / 64
        wc = m.get('SQ_WAVE_CYCLES', 0) or 1
        print(f'{k[:58]:58s} {dur[k]:8.3f} {cyc / dur[k] / 1e6:6.3f} {m.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / (1024 * cyc):9.3f} '
              f'{m.get("SQ_WAIT_ANY", 0) / wc:6.3f} {m.get("SQ_WAIT_INST_ANY", 0) / wc:11.3f} {m.get("SQ_ACTIVE_INST_ANY", 0) / wc:6.3f} '
              f'{((m.get("SQ_INSTS_VALU", 0) - n_mfma) / n_mfma) if n_mfma else 0:9.2f} {(m.get("SQ_INSTS_LDS", 0) / n_mfma) if n_mfma else 0:13.2f}')
