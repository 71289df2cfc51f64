timeout 600 python -m pytest tests -m gpu -x -q 2>&1 | tail -1
for i in 1 2; do python bench.py --no-cpu-baseline 2>&1 | tail -1 | python -c "import json,sys; d=json.load(sys.stdin); print(d[\"value\"], d[\"frames_bitwise_reproducible\"], d[\"roofline\"][\"family_ms_per_step\"][\"upfirdn2d\"])"; done
