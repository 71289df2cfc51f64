tag=r04
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
tail -5 $out/traffic_summary.log
rm -rf $out/cf $out/cw $out/pmc_calib $out/pf $out/pw
cp $out/${tag}_traffic_pmc.json profiles/
python bench.py --no-extras --no-cpu-baseline > $out/${tag}_bench_noextras.json 2>/dev/null; python - <<'PY'
import json
d=json.loads(open('gpurun_out/art/r04_bench_noextras.json').read().strip().splitlines()[-1])
r=d['roofline']; print(d['value'], r['frac'], r['traffic'], r['algorithmic_bytes_per_launch'], r['traffic_source'])
PY
