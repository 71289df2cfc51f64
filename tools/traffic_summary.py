#!/usr/bin/env python3
"""HBM traffic per launch of the roofline kernel family from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE), calibrated on
kernels that move a known byte count (tools/pmc_calib.hip, same two passes).
Usage: traffic_summary.py fetch.db write.db out.json [calib_fetch.db calib_write.db]
(MI355X_MICROARCH.md §HBM: separate passes; FETCH_SIZE / WRITE_SIZE come in KiB; gfx950 FETCH_SIZE under-reports wide reads)"""
import collections, json, re, sqlite3, sys

FAMILY = re.compile(r'^(void )?conv2d(_[a-z0-9_]+)?_bf16x3(_pair)?_kernel')      # every 3x3 split-bf16 kernel, whatever its variant suffix (not conv1x1_* / conv16_splitk_* / rgb_combine)
LAUNCHES_PER_STEP = 69          # bench.py configs[1], force_fp32 route (roofline.launches_per_step): 67 until round 4, + the two few-pixel stride-2 layers that were fp32-MFMA launches
CALIB_BYTES = float(1 << 30)


def per_kernel(db, counter):
    con = sqlite3.connect(db)
    cols = [r[1] for r in con.execute("pragma table_info('counters_collection')")]
    ix = {c: i for i, c in enumerate(cols)}
    agg = collections.defaultdict(list)
    for r in con.execute("select * from counters_collection"):
        if r[ix['counter_name']] != counter:
            continue
        name = (r[ix['kernel_name']] if 'kernel_name' in ix else r[ix['name']]).split('(')[0]
        agg[name].append(r[ix['value']])
    return agg


f, w = per_kernel(sys.argv[1], 'FETCH_SIZE'), per_kernel(sys.argv[2], 'WRITE_SIZE')
sel = lambda d: {k: v for k, v in d.items() if FAMILY.match(k)}
f, w = sel(f), sel(w)
nl, nlw = sum(len(v) for v in f.values()), sum(len(v) for v in w.values())
assert nl > 0 and (nl % LAUNCHES_PER_STEP == 0 or nl % 67 == 0 or nl % 60 == 0) and nlw == nl, f'family launches counted: fetch pass {nl}, write pass {nlw} — expected a multiple of {LAUNCHES_PER_STEP} (run bench.py with --no-extras --lanes 1)'
fetch_kb = sum(sum(v) for v in f.values()) / nl
write_kb = sum(sum(v) for v in w.values()) / nl
out = {
    'source': 'rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (two separate passes, --kernel-trace only) on `bench.py --no-extras --lanes 1 --steps 2 --warmup 1`, '
              'averaged over ALL launches of the 3x3 split-bf16 conv family (bench.py\'s roofline family; kernel names below)',
    'per_kernel_avg_KB': {'FETCH_SIZE': {k: sum(v) / len(v) for k, v in f.items()}, 'WRITE_SIZE': {k: sum(v) / len(v) for k, v in w.items()}},
    'per_kernel_launches': {k: len(v) for k, v in f.items()},
    'launches_counted': nl, 'steps_counted': nl // (LAUNCHES_PER_STEP if nl % LAUNCHES_PER_STEP == 0 else (67 if nl % 67 == 0 else 60)),
    'fetch_KB_per_launch_raw': fetch_kb, 'write_KB_per_launch_raw': write_kb,
    'traffic_bytes_per_launch_raw': (fetch_kb + write_kb) * 1024.0,
}
if len(sys.argv) > 5:
    cf, cw = per_kernel(sys.argv[4], 'FETCH_SIZE'), per_kernel(sys.argv[5], 'WRITE_SIZE')
    cal = {}
    for k in ('calib_copy16_kernel', 'calib_copy4_kernel', 'calib_dma16_kernel'):
        fk = [v for n, vs in cf.items() if k in n for v in vs]
        wk = [v for n, vs in cw.items() if k in n for v in vs]
        cal[k] = {'FETCH_SIZE_KB': sum(fk) / len(fk), 'WRITE_SIZE_KB': sum(wk) / len(wk), 'known_bytes_each_way': CALIB_BYTES,
                  'fetch_counter_per_byte': sum(fk) / len(fk) * 1024.0 / CALIB_BYTES, 'write_counter_per_byte': sum(wk) / len(wk) * 1024.0 / CALIB_BYTES}
    out['calibration'] = cal
    # the pre-split kernels (79 % of the family's time) fetch by LDS-DMA 16 B/lane and store 16 B/lane; the register-staged ones gather 4 B/lane
    dma, c4, c16 = cal['calib_dma16_kernel'], cal['calib_copy4_kernel'], cal['calib_copy16_kernel']
    tot_f = tot_w = 0.0
    for k, v in f.items():
        staged_by_dma = '_ps' in k
        kf = (dma if staged_by_dma else c4)['fetch_counter_per_byte']
        tot_f += sum(v) * 1024.0 / kf
    for k, v in w.items():
        tot_w += sum(v) * 1024.0 / c16['write_counter_per_byte']
    out['traffic_bytes_per_launch_calibrated'] = (tot_f + tot_w) / nl
    out['fetch_bytes_per_launch_calibrated'] = tot_f / nl
    out['write_bytes_per_launch_calibrated'] = tot_w / nl
    out['calibration_note'] = ('counter / known bytes of tools/pmc_calib.hip (1 GiB each way, beyond the Infinity Cache): FETCH_SIZE of the LDS-DMA kernels is divided '
                               'by the dma16 factor, of the register-staged kernels by the copy4 factor, WRITE_SIZE by the copy16 factor')
json.dump(out, open(sys.argv[3], 'w'), indent=1)
print(json.dumps(out, indent=1))
