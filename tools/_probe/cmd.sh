cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 600 python tools/step_ab.py --attr small_bias_pass_on_main 2>&1 | grep -v amdgpu | tail -4
timeout 900 python -m pytest tests -m gpu -q -x -k "bias or ones_channel or gradients or round3" 2>&1 | grep -v amdgpu | tail -3
