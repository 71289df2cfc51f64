#!/bin/bash
# GPU box: rocprofv3 kernel stats of 40 eager single-frame forwards (the scripts' call pattern), float32 route and default route.
tag=${1:-r05}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/art; mkdir -p $out
for mode in "" "--default-route"; do
  sfx=$([ -z "$mode" ] && echo "" || echo "_default_route")
  rocprofv3 --kernel-trace --stats -d $out/kb1 -o r -- python tools/batch1_frames.py 40 $mode > $out/${tag}_batch1${sfx}.log 2> $out/kb1.err
  python tools/rocpd_summary.py $(find $out/kb1 -name "*.db" | head -1) $out/${tag}_kernel_stats_batch1${sfx}.csv > /dev/null
  rm -rf $out/kb1
  python tools/batch1_frames.py 40 $mode | tail -1
  tail -1 $out/${tag}_batch1${sfx}.log
done
