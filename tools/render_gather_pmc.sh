#!/bin/bash
# GPU box: PMC passes over the renderer alone (tools/render_only.py, tuning build) for round 5's gather (N3D_RENDER_GATHER=0: a lane fetches 64 bytes of its own sample's texels)
# and the coalesced one (=1: eight adjacent lanes fetch one 128-byte texel).  tools/build_variant.sh tune render.hip -DN3D_TUNING first.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export N3D_LIB=tools/probe/libn3d_tune.so
out=gpurun_out/r06_render_gather_pmc.txt; : > $out
for mode in 0 1; do
  export N3D_RENDER_GATHER=$mode
  echo "## N3D_RENDER_GATHER=$mode" >> $out
  python tools/render_only.py --iters 20 2>&1 | grep "render (bounds" >> $out
  i=0
  for grp in "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum" \
             "SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_VMEM_RD"; do
    i=$((i+1))
    rocprofv3 --kernel-trace --pmc $grp -d gpurun_out/rpmc_$i -o r -- python tools/render_only.py --iters 3 > /dev/null 2>gpurun_out/rpmc_$i.err
    db=$(find gpurun_out/rpmc_$i -name "*.db" | head -1)
    python tools/pmc_summary.py $db render_rays >> $out 2>&1
    rm -rf gpurun_out/rpmc_$i gpurun_out/rpmc_$i.err
  done
done
cat $out
