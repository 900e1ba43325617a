cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 600 python tools/e2e_train_throughput.py --from-audio --steps 100 2>&1 | grep -v amdgpu | tail -4
