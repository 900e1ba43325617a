cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 2400 python -m pytest tests -m gpu -q 2>&1 | grep -v amdgpu | tail -15 > gpurun_out/r03_gpu_tests.txt
grep -n "passed\|failed" gpurun_out/r03_gpu_tests.txt
bash tools/profile_round.sh r03k
cat gpurun_out/r03k_step_time_by_dtype.txt
