#!/usr/bin/env python3
"""Print VGPR/AGPR/SGPR/LDS/occupancy per kernel of one .hip file (hipcc -Rpass-analysis=kernel-resource-usage)."""
import re, subprocess, sys
src = sys.argv[1]
extra = sys.argv[2:]
r = subprocess.run(['/opt/rocm/bin/hipcc', '-O3', '-std=c++17', '-fPIC', '--offload-arch=gfx950',
                    '-Rpass-analysis=kernel-resource-usage', '-c', src, '-o', '/dev/null'] + extra, capture_output=True, text=True)
cur = None
rows = {}
for line in r.stderr.splitlines():
    m = re.search(r'Function Name: (\S+)', line)
    if m:
        cur = subprocess.run(['c++filt', m.group(1)], capture_output=True, text=True).stdout.strip()
        rows[cur] = {}
        continue
    m = re.search(r'remark:\s+([A-Za-z][^:]*): (\d+)', line)
    if m and cur:
        rows[cur][m.group(1).strip()] = int(m.group(2))
for k, v in rows.items():
    name = re.sub(r'\(.*', '', k)[:90]
    print(f"{name:90s} vgpr={v.get('VGPRs')} agpr={v.get('AGPRs')} sgpr={v.get('SGPRs')} spillS={v.get('SGPRs Spill')} "
          f"spillV={v.get('VGPRs Spill')} scratch={v.get('ScratchSize [bytes/lane]')} occ={v.get('Occupancy [waves/SIMD]')} lds={v.get('LDS Size [bytes/block]')}")
if r.returncode:
    print(r.stderr[-3000:])
