#!/bin/bash
# GPU box: every measured artefact of round 6 in one call (results under gpurun_out/art/): tools/refresh_profiles.sh r06 (bench line, rocprofv3 kernel stats of both routes, calibrated
# HBM traffic, matrix-pipe busy), batch-1 kernel stats, layer traces at batch 4 / 1, a soak run, the renderer's PMC passes on the final kernels.
cd $GRAFT_REPO_ROOT
tools/refresh_profiles.sh r06 > gpurun_out/refresh_r06.log 2>&1
tools/batch1_stats.sh r06 >> gpurun_out/refresh_r06.log 2>&1
python tools/layer_trace.py > gpurun_out/art/r06_layer_trace.txt 2>&1
python tools/layer_trace.py --batch 1 > gpurun_out/art/r06_layer_trace_batch1.txt 2>&1
{ echo "# tools/soak.py on the round-6 build: forwards of the same inputs round-robin on three HIP streams, every image compared bit for bit with the first"; python tools/soak.py --steps 600 2>&1 | grep -v amdgpu; python tools/soak.py --steps 600 --fp16 2>&1 | grep -v amdgpu; } > gpurun_out/art/r06_soak.txt
tools/render_gather_pmc.sh > /dev/null 2>&1; cp gpurun_out/r06_render_gather_pmc.txt gpurun_out/art/r06_render_pmc_final.txt
tail -5 gpurun_out/refresh_r06.log; cat gpurun_out/art/r06_soak.txt; head -c 600 gpurun_out/art/r06_bench.json
