for s in "512 256 64 64 2" "256 128 128 128 2" "256 128 256 256 2" "512 512 32 32 2"; do python tools/conv16_sweep.py $s 2>/dev/null | grep -E "auto|ksplit 1:"; done
timeout 600 python -m pytest tests -m gpu -x -q 2>&1 | tail -2
for i in 1 2; do timeout 300 python bench.py --no-cpu-baseline 2>&1 | tail -1 | python -c "import json,sys; d=json.load(sys.stdin); print(d[\"value\"], d[\"roofline\"][\"family_ms_per_step\"])"; done
