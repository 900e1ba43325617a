cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python tools/batch_sweep.py 2>&1 | grep -v amdgpu | tee gpurun_out/r03_batch_sweep.txt | tail -12
