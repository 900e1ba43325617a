cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 600 python tools/power_probe.py 2>&1 | grep -v amdgpu | tee gpurun_out/r03_power_probe.txt | tail -20
