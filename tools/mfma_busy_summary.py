#!/usr/bin/env python3
"""Matrix-pipe utilisation of the 3x3 split-bf16 conv family from one rocprofv3 --pmc pass
(SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE).  MI355X_MICROARCH.md: SQ_VALU_MFMA_BUSY_CYCLES counts cycles (32 per
v_mfma_f32_32x32x16_bf16), summed over the chip's 1024 SIMDs; GRBM_GUI_ACTIVE comes back summed over the 8 XCDs (checked: /8 it
equals kernel time x ~1.9-2.0 GHz, the clock the chip sustains under this load).  Raw counters are kept next to the ratios.
Usage: mfma_busy_summary.py results.db out.json [kernel_stats.csv]"""
import collections, json, sqlite3, sys
con = sqlite3.connect(sys.argv[1])
cols = [r[1] for r in con.execute("pragma table_info('counters_collection')")]
ix = {c: i for i, c in enumerate(cols)}
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in con.execute("select * from counters_collection"):
    name = (r[ix['kernel_name']] if 'kernel_name' in ix else r[ix['name']]).split('(')[0]
    if (('bf16x3' in name and 'conv1x1' not in name and 'prep' not in name and 'splitk' not in name) or ('h8_f16' in name and name.startswith(('conv2d', 'void conv2d')))):
        agg[name][r[ix['counter_name']]].append(r[ix['value']])
out = {'source': 'rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE on `bench.py --lanes 1 --steps 2 --warmup 1` (+ --sr-fp16 for the float16-block kernels)',
       'note': 'mfma_busy = SQ_VALU_MFMA_BUSY_CYCLES / ((GRBM_GUI_ACTIVE / 8 XCDs) * 1024 SIMDs): fraction of SIMD-cycles the matrix pipe was '
               'busy while the kernel ran; cycles_per_launch = GRBM_GUI_ACTIVE / 8',
       'kernels': {}}
for k, d in sorted(agg.items()):
    busy, act = d.get('SQ_VALU_MFMA_BUSY_CYCLES', []), d.get('GRBM_GUI_ACTIVE', [])
    if not busy or not act:
        continue
    out['kernels'][k] = {'launches': len(busy), 'SQ_VALU_MFMA_BUSY_CYCLES_avg': sum(busy) / len(busy), 'GRBM_GUI_ACTIVE_avg': sum(act) / len(act),
                         'cycles_per_launch': sum(act) / len(act) / 8.0, 'mfma_busy': sum(busy) / (sum(act) / 8.0 * 1024.0)}
json.dump(out, open(sys.argv[2], 'w'), indent=1)
print(json.dumps(out, indent=1))
