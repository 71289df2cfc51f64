python -m pytest tests/test_ops_gpu.py -m gpu -x -q -k "bf16x3 or row_pitch or synthesis_layer" 2>&1 | tail -3
for s in "512 512 16 16 0" "1024 512 16 16 0" "512 512 8 8 0" "1024 512 8 8 0" "512 512 4 4 0" "512 512 4 4 2"; do python tools/conv16_sweep.py $s 2>/dev/null; done
