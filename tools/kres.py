#!/usr/bin/env python3
"""Compile one .hip file for gfx950 and print a per-kernel resource table
(VGPRs, AGPRs, spills, SGPRs, LDS, occupancy) from -Rpass-analysis=kernel-resource-usage."""
import re, subprocess, sys, os
src = sys.argv[1]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
cmd = ['hipcc', '-O3', '--offload-arch=gfx950', '-fPIC', '-I', os.path.join(root, 'include'),
       '-c', src, '-o', '/dev/null', '-Rpass-analysis=kernel-resource-usage'] + sys.argv[2:]
p = subprocess.run(cmd, capture_output=True, text=True)
rows, cur = [], None
for line in p.stderr.splitlines():
    if 'error' in line or 'warning' in line:
        print(line)
    m = re.search(r'remark: +(.*?): (.*?) \[-Rpass', line)
    if not m:
        continue
    k, v = m.group(1).strip(), m.group(2).strip()
    if k == 'Function Name':
        cur = {'name': subprocess.run(['c++filt', v], capture_output=True, text=True).stdout.strip()[:70]}
        rows.append(cur)
    elif cur is not None:
        cur[k] = v
print(f"{'kernel':70s} {'VGPR':>5} {'AGPR':>5} {'vspill':>6} {'SGPR':>5} {'sspill':>6} {'LDS':>7} {'occ':>4} {'scratch':>7}")
for r in rows:
    print(f"{r['name']:70s} {r.get('VGPRs','?'):>5} {r.get('AGPRs','?'):>5} {r.get('VGPRs Spill','?'):>6} "
          f"{r.get('TotalSGPRs','?'):>5} {r.get('SGPRs Spill','?'):>6} {r.get('LDS Size [bytes/block]','?'):>7} "
          f"{r.get('Occupancy [waves/SIMD]','?'):>4} {r.get('ScratchSize [bytes/lane]','?'):>7}")
sys.exit(p.returncode)
