cd /root/repo
mkdir -p gpurun_out
export RASTER_CLASSIFY=1
( echo "=== plain-load rasteriser (RASTER_VARIANT 64 = 0 + raw-word dump) beside the real convolution kernel ==="; timeout 300 tools/probe/raster_repro tools/probe/libraster_v64.so next3d_amd/libn3d.so tools/probe/raster_inputs.bin 24
  echo "=== shipped loads (RASTER_VARIANT 79 = 15 + raw-word dump) ==="; timeout 300 tools/probe/raster_repro tools/probe/libraster_v79.so next3d_amd/libn3d.so tools/probe/raster_inputs.bin 24 ) > gpurun_out/r5_raster_classified.txt 2>&1
unset RASTER_CLASSIFY
timeout 600 python tools/render_coresidency.py --launches 20000 > gpurun_out/r5_render_coresidency.txt 2>&1
timeout 900 bash tools/render_pmc.sh > /dev/null 2>&1; cp gpurun_out/render_pmc.txt gpurun_out/r5_render_pmc.txt
timeout 600 python -m pytest tests -m gpu -q 2>&1 | tail -15 > gpurun_out/r5_tests4.txt
timeout 900 python bench.py > gpurun_out/r5_bench1.json 2> gpurun_out/r5_bench1.err
tail -40 gpurun_out/r5_raster_classified.txt; cat gpurun_out/r5_render_coresidency.txt | tail -5; tail -5 gpurun_out/r5_tests4.txt; head -c 1500 gpurun_out/r5_bench1.json
