# scratch command file for `gpurun -- 'bash tools/_probe/cmd.sh'` (edited per experiment; the last one run on the GPU box)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 2400 python -m pytest tests -m gpu -q 2>&1 | grep -v amdgpu | tail -5
