python tools/run_one.py --kind wgrad_multi --layer striding_conv --cfg 0 --reps 50 2>&1 | grep -v amdgpu
python tools/run_one.py --kind wgrad_multi --layer striding_conv --cfg 1 --reps 50 2>&1 | grep -v amdgpu
python tools/run_one.py --kind wgrad_grouped --layer inner_conv_1 --reps 50 2>&1 | grep -v amdgpu
python tools/run_one.py --kind wgrad --layer striding_conv --reps 50 2>&1 | grep -v amdgpu
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r03i_prof -o p -- python tools/run_one.py --kind wgrad_multi --layer striding_conv --cfg 0 --reps 30 > /dev/null 2>&1
grep -h "multi\|reduce" gpurun_out/r03i_prof/*kernel_stats.csv | cut -c1-200
