python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "every_wgrad_tile" 2>&1 | tail -3
for L in big_conv_2 big_conv_1; do for cfg in 0 2884; do python tools/run_one.py --kind wgrad --layer $L --cfg $cfg --reps 40 2>&1 | grep -v amdgpu; done; done
for cfg in 0 2884; do python tools/run_one.py --kind wgrad_grouped --layer inner_conv_1 --cfg $cfg --reps 40 2>&1 | grep -v amdgpu; done
