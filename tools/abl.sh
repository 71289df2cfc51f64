timeout 600 python -m pytest tests -m gpu -x -q 2>&1 | tail -1
for i in 1 2; do
for L in gpurun_ab_prev.so next3d_amd/libn3d.so; do echo $L; N3D_LIB=$PWD/$L timeout 300 python bench.py --no-cpu-baseline 2>&1 | tail -1 | python -c "import json,sys; d=json.load(sys.stdin); print(d[\"value\"], d[\"roofline\"][\"family_ms_per_step\"][\"conv1x1_bf16x3\"], d[\"roofline\"][\"family_ms_per_step\"][\"misc\"])"; done; done
