cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests/test_gpu_round3.py -m gpu -q -k "learnt" 2>&1 | grep -v amdgpu | grep "^E  \|passed\|failed" | cut -c1-300 | tail -12
