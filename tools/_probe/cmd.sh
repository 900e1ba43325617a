cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests -m gpu -q -x -k "ctc or lattice or repair or learnt or long_labels or full_length" 2>&1 | grep -v amdgpu | grep "^E  \|passed\|failed" | cut -c1-250 | tail -5
