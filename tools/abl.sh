python -m pytest tests/test_ops_gpu.py -m gpu -x -q -k "stride2 or row_pitch" 2>&1 | tail -5
for s in "128 256 257 257 1" "256 512 129 129 1" "512 512 65 65 1"; do python tools/conv16_sweep.py $s 2>/dev/null; done
