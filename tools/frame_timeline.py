#!/usr/bin/env python3
"""Timeline of ONE steady-state frame from a rocprofv3 kernel trace (rocpd SQLite) of tools/batch1_frames.py: every dispatch in start order with
its queue, start offset, duration and the idle time of ITS queue in front of it — where the device-side chain of a batch-1 frame goes.
    rocprofv3 --kernel-trace -d gpurun_out/kt -o r -- python tools/batch1_frames.py 12 --default-route
    python tools/frame_timeline.py $(find gpurun_out/kt -name '*.db' | head -1) [frame index from the end, default 2]"""
import re, sqlite3, sys
db = sys.argv[1]
back = int(sys.argv[2]) if len(sys.argv) > 2 else 2
con = sqlite3.connect(db)
cols = [r[1] for r in con.execute("pragma table_info(kernels)")]
print('# columns:', cols)
qcol = 'queue_id' if 'queue_id' in cols else ('stream_id' if 'stream_id' in cols else None)
rows = con.execute(f"select name, start, end, {qcol or '0'} from kernels order by start").fetchall()
marks = [k for k, r in enumerate(rows) if 'normalize_2nd_moment_kernel<double>' in r[0] or 'normalize_2nd_moment_kernel<float>' in r[0]]
# a frame starts at the first normalize launch of mapping (two per frame: z and c): take every second mark
starts = marks[::2]
a, b = starts[-back - 1], starts[-back]
frame = rows[a:b]
t0 = frame[0][1]
last_end = {}
busy_union, cur_end = 0, t0
print(f'# frame of {len(frame)} dispatches, {(rows[b][1] - t0) / 1e3:.1f} us from its first dispatch to the next frame\'s first')
tot = {}
for name, s, e, q in frame:
    short = re.sub(r'\(.*', '', name)
    short = re.sub(r'^void ', '', short)[:46]
    gap = (s - last_end[q]) / 1e3 if q in last_end else 0.0
    last_end[q] = e
    if s > cur_end:
        cur_end = s
    if e > cur_end:
        busy_union += e - cur_end; cur_end = e
    tot[short] = tot.get(short, 0) + (e - s)
    print(f'{(s - t0) / 1e3:8.1f} us  q{q}  {(e - s) / 1e3:7.1f} us  (queue idle before: {gap:6.1f})  {short}')
print(f'# device busy (union over queues): {busy_union / 1e3:.1f} us; sum of kernel durations {sum(e - s for _, s, e, _ in frame) / 1e3:.1f} us')
for k, v in sorted(tot.items(), key=lambda kv: -kv[1])[:25]:
    print(f'#   {v / 1e3:8.1f} us  {k}')
