#!/bin/bash
# GPU box: regenerate the artefacts kept under profiles/ (bench line, rocprofv3 kernel-trace summary, PMC traffic).
tag=${1:-r01}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/art
if [ "$2" != "lite" ]; then python bench.py > gpurun_out/art/${tag}_bench.json 2> gpurun_out/art/bench.err; fi
rocprofv3 --kernel-trace --stats -d gpurun_out/art/kt -o r -- python bench.py --no-cpu-baseline > gpurun_out/art/${tag}_bench_under_rocprofv3.json 2> gpurun_out/art/kt.err
python tools/rocpd_summary.py $(find gpurun_out/art/kt -name "*.db" | head -1) gpurun_out/art/${tag}_kernel_stats.csv > /dev/null
if [ "$2" = "lite" ]; then      # kernel trace first, bench line last, no PMC passes (the committed traffic file stays): fits a short GPU slot
    rm -rf gpurun_out/art/kt
    python bench.py > gpurun_out/art/${tag}_bench.json 2> gpurun_out/art/bench.err
    head -c 1500 gpurun_out/art/${tag}_bench.json; echo; head -12 gpurun_out/art/${tag}_kernel_stats.csv
    exit 0
fi
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d gpurun_out/art/pf -o r -- python bench.py --no-cpu-baseline --no-roofline --steps 2 --warmup 1 > /dev/null 2> gpurun_out/art/pf.err
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d gpurun_out/art/pw -o r -- python bench.py --no-cpu-baseline --no-roofline --steps 2 --warmup 1 > /dev/null 2> gpurun_out/art/pw.err
python tools/traffic_summary.py $(find gpurun_out/art/pf -name "*.db" | head -1) $(find gpurun_out/art/pw -name "*.db" | head -1) gpurun_out/art/${tag}_traffic_pmc.json > /dev/null
rm -rf gpurun_out/art/kt gpurun_out/art/pf gpurun_out/art/pw
head -c 1500 gpurun_out/art/${tag}_bench.json; echo; head -12 gpurun_out/art/${tag}_kernel_stats.csv
