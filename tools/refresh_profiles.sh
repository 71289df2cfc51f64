#!/bin/bash
# GPU box: regenerate the artefacts kept under profiles/ for one round: bench line, rocprofv3 kernel-trace summary of the same
# command, PMC traffic (FETCH_SIZE / WRITE_SIZE, separate passes) and matrix-pipe busy counters of the 3x3 conv family.
# usage: tools/refresh_profiles.sh r02 [lite]     (lite: no PMC passes)
tag=${1:-r03}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/art; mkdir -p $out
rocprofv3 --kernel-trace --stats -d $out/kt -o r -- python bench.py --no-cpu-baseline --no-extras > $out/${tag}_bench_under_rocprofv3.json 2> $out/kt.err
python tools/rocpd_summary.py $(find $out/kt -name "*.db" | head -1) $out/${tag}_kernel_stats.csv > /dev/null
rm -rf $out/kt
# the scripts' default route (float16 super-resolution blocks on the f16 matrix cores): kernel trace of the same command with --sr-fp16
rocprofv3 --kernel-trace --stats -d $out/kt16 -o r -- python bench.py --no-cpu-baseline --no-extras --sr-fp16 > $out/${tag}_bench_sr_fp16_under_rocprofv3.json 2> $out/kt16.err
python tools/rocpd_summary.py $(find $out/kt16 -name "*.db" | head -1) $out/${tag}_kernel_stats_sr_fp16.csv > /dev/null
rm -rf $out/kt16
if [ "$2" != "lite" ]; then
  P="python bench.py --no-cpu-baseline --no-roofline --no-extras --lanes 1 --steps 2 --warmup 1 --prewarm-seconds 0"
  rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $out/pf -o r -- $P > /dev/null 2> $out/pf.err
  rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $out/pw -o r -- $P > /dev/null 2> $out/pw.err
  hipcc -O3 --offload-arch=gfx950 tools/pmc_calib.hip -o $out/pmc_calib
  rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $out/cf -o r -- $out/pmc_calib > $out/calib.log 2> $out/cf.err
  rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $out/cw -o r -- $out/pmc_calib >> $out/calib.log 2> $out/cw.err
  python tools/traffic_summary.py $(find $out/pf -name "*.db" | head -1) $(find $out/pw -name "*.db" | head -1) $out/${tag}_traffic_pmc.json \
         $(find $out/cf -name "*.db" | head -1) $(find $out/cw -name "*.db" | head -1) > $out/traffic_summary.log 2>&1
  rm -rf $out/cf $out/cw $out/pmc_calib
  rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE -d $out/pm -o r -- $P > /dev/null 2> $out/pm.err
  python tools/mfma_busy_summary.py $(find $out/pm -name "*.db" | head -1) $out/${tag}_mfma_busy_pmc.json > /dev/null
  rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE -d $out/pm16 -o r -- $P --sr-fp16 > /dev/null 2> $out/pm16.err
  python tools/mfma_busy_summary.py $(find $out/pm16 -name "*.db" | head -1) $out/${tag}_mfma_busy_sr_fp16_pmc.json > /dev/null
  rm -rf $out/pf $out/pw $out/pm $out/pm16
  cp $out/${tag}_traffic_pmc.json $out/${tag}_mfma_busy_pmc.json $out/${tag}_mfma_busy_sr_fp16_pmc.json profiles/     # the bench line below cites the fresh traffic profile
fi
python bench.py > $out/${tag}_bench.json 2> $out/bench.err
head -c 1200 $out/${tag}_bench.json; echo; head -14 $out/${tag}_kernel_stats.csv; ls -la $out
