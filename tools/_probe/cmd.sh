cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests -m gpu -q -x -k "pack or input or config5 or long_form or full_length or stack or golden" 2>&1 | grep -v amdgpu | tail -4
timeout 600 python bench.py --config 5 --no-cpu-baseline > gpurun_out/c5.json; python - <<'PY'
import json
d=json.loads(open('gpurun_out/c5.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['step_mfma_frac']); print(json.dumps(d['kernels']['per_launch_ms']))
PY
