# GPU box (round 4): transposed kernels, adjacent output phases stored back to back vs the previous order (N3D_LIB=tools/probe/libn3d_prev.so = a build of the previous
# commit), one launch at a time (profiles/r04_up_store_order_ab.txt; the in-model re-check used tools/layer_trace.py and bench.py with the same two libraries).
cd /root/repo
mkdir -p gpurun_out
for rep in 1 2; do
echo "== previous library"; N3D_LIB=tools/probe/libn3d_prev.so python tools/ps_splitk_bench.py 2>/dev/null | tail -7
echo "== stores reordered"; python tools/ps_splitk_bench.py 2>/dev/null | tail -7
done
echo "== f16 previous"; N3D_LIB=tools/probe/libn3d_prev.so python tools/f16_bench.py 2>/dev/null | grep -E "^---|transposed"
echo "== f16 reordered"; python tools/f16_bench.py 2>/dev/null | grep -E "^---|transposed"
python -m pytest tests/test_ops_gpu.py tests/test_f16_gpu.py -q -m gpu -k "transposed or up_ or presplit or thin" 2>&1 | tail -2
