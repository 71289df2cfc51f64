#!/bin/bash
# usage: tools/pmc_run.sh <kernel-substring> "<COUNTER ...>" ["<COUNTER ...>" ...]  -- one rocprofv3 --pmc pass per counter group
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
sub=$1; shift
i=0
for grp in "$@"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $grp -d gpurun_out/pmc_$i -o r -- python bench.py --no-cpu-baseline --no-roofline --steps 2 --warmup 1 > /dev/null 2>gpurun_out/pmc_$i.err
  db=$(find gpurun_out/pmc_$i -name "*.db" | head -1)
  python tools/pmc_summary.py $db $sub
done
