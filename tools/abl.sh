timeout 500 python -m pytest tests/test_networks_gpu.py tests/test_generator_gpu.py -m gpu -x -q 2>&1 | tail -1
python - <<'PY'
import json, subprocess, sys, os
for mode in ('copy', 'inplace', 'copy', 'inplace'):
    code = "import next3d_amd.networks as n; n._CAT_COPY = %s; import runpy, sys; sys.argv=['bench.py','--no-cpu-baseline','--no-roofline']; runpy.run_path('bench.py', run_name='__main__')" % (mode == 'copy')
    out = subprocess.run([sys.executable, '-c', code], capture_output=True, text=True).stdout.strip().splitlines()[-1]
    print(mode, json.loads(out)['value'])
PY
