#!/bin/bash
# GPU box: only the HBM-traffic passes of tools/refresh_profiles.sh (FETCH_SIZE / WRITE_SIZE of the 3x3 family + the calibration kernels) and the final bench line.
tag=${1:-r05}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/art; mkdir -p $out
P="python bench.py --no-cpu-baseline --no-roofline --no-extras --lanes 1 --steps 2 --warmup 1 --prewarm-seconds 0"
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $out/pf -o r -- $P > /dev/null 2> $out/pf.err
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $out/pw -o r -- $P > /dev/null 2> $out/pw.err
hipcc -O3 --offload-arch=gfx950 tools/pmc_calib.hip -o $out/pmc_calib
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $out/cf -o r -- $out/pmc_calib > $out/calib.log 2> $out/cf.err
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $out/cw -o r -- $out/pmc_calib >> $out/calib.log 2> $out/cw.err
python tools/traffic_summary.py $(find $out/pf -name "*.db" | head -1) $(find $out/pw -name "*.db" | head -1) $out/${tag}_traffic_pmc.json \
       $(find $out/cf -name "*.db" | head -1) $(find $out/cw -name "*.db" | head -1) > $out/traffic_summary.log 2>&1
rm -rf $out/cf $out/cw $out/pmc_calib $out/pf $out/pw
cp $out/${tag}_traffic_pmc.json profiles/
python bench.py > $out/${tag}_bench.json 2> $out/bench.err
tail -12 $out/traffic_summary.log; head -c 300 $out/${tag}_bench.json
