cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_spectrogram.py -m gpu -q 2>&1 | grep -v amdgpu | tail -6
timeout 300 python tools/_probe/front_end.py 2>&1 | grep -v amdgpu | tail -2
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/fe_prof -o p -- python tools/_probe/front_end.py > /dev/null 2>&1
python - <<'PY'
import csv
for r in list(csv.reader(open('gpurun_out/fe_prof/p_kernel_stats.csv')))[:6]:
    print(r[0][:90], r[1], r[3][:9], r[4])
PY
