cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_round3.py tests/test_gpu_parity.py -m gpu -q -k "output_softmax or permutation or bf16_matches" 2>&1 | grep -v amdgpu | grep "^E  \|passed\|failed" | cut -c1-250 | tail -8
