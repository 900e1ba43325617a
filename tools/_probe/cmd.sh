cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 300 python tools/ctc_time.py --aligned 20 2>&1 | grep -v amdgpu | grep "float" | cut -c1-120
