timeout 600 python -m pytest tests/test_generator_gpu.py -m gpu -x -q 2>&1 | tail -2
for i in 1 2; do for f in 0 1; do echo "raster_side=$f"; N3D_OVERLAP_RASTER=$f timeout 300 python bench.py --no-cpu-baseline --no-roofline 2>&1 | tail -1 | python -c "import json,sys; d=json.load(sys.stdin); print(d[\"value\"])"; done; done
