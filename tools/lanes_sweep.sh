#!/bin/bash
# GPU box: the headline loop of bench.py for several lane counts (HIP streams the steps alternate over), same box, back to back.
for l in ${@:-3 2 4 5 6 3}; do
  python bench.py --no-cpu-baseline --no-roofline --no-extras --lanes $l --steps 40 --warmup 8 2>/dev/null > /tmp/l.json
  python - "$l" <<'PY'
import json, sys
d = json.load(open('/tmp/l.json'))
print(f"lanes {sys.argv[1]}: {d['value']:.1f} frames/s, {d['ms_per_step']:.3f} ms per step")
PY
done
