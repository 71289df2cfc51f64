#!/usr/bin/env python3
"""Summarise a rocprofv3 rocpd SQLite result (kernel-trace) into the per-kernel stats table rocprofv3 --stats would
print: name, calls, total ms, avg us, min/max us, % of GPU time.  Usage: rocpd_summary.py results.db [out.csv]"""
import re
import sqlite3
import sys

db, out = sys.argv[1], (sys.argv[2] if len(sys.argv) > 2 else None)
con = sqlite3.connect(db)
rows = con.execute("select name, count(*), sum(duration), avg(duration), min(duration), max(duration), "
                   "max(vgpr_count), max(accum_vgpr_count), max(lds_size) from kernels group by name order by sum(duration) desc").fetchall()
total = sum(r[2] for r in rows)
lines = ['name,calls,total_ms,avg_us,min_us,max_us,pct,vgpr,agpr,lds_bytes']
for name, calls, tot, avg, mn, mx, vg, ag, lds in rows:
    short = re.sub(r'\(.*', '', name).replace(',', ';')
    lines.append(f'"{short}",{calls},{tot / 1e6:.3f},{avg / 1e3:.2f},{mn / 1e3:.2f},{mx / 1e3:.2f},{100 * tot / total:.2f},{vg},{ag},{lds}')
text = '\n'.join(lines)
print(text)
if out:
    open(out, 'w').write(text + '\n')
