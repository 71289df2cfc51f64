timeout 600 python -m pytest tests/test_ops_gpu.py -m gpu -x -q -k "up_bf16x3 or synthesis or row_pitch" 2>&1 | tail -2
for s in "256 128 256 256 2" "512 256 64 64 2"; do for L in gpurun_ab_prev.so next3d_amd/libn3d.so; do N3D_LIB=$PWD/$L python tools/conv16_sweep.py $s 2>/dev/null | grep "ksplit 1:"; done; done
for i in 1 2; do
for L in gpurun_ab_prev.so next3d_amd/libn3d.so; do echo $L; N3D_LIB=$PWD/$L timeout 300 python bench.py --no-cpu-baseline 2>&1 | tail -1 | python -c "import json,sys; d=json.load(sys.stdin); print(d[\"value\"], d[\"roofline\"][\"family_ms_per_step\"][\"conv2d_bf16x3\"])"; done; done
