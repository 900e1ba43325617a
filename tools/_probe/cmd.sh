cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python tools/_probe/garbage_probe.py 2>&1 | grep -v amdgpu | tail -20
