for r in 1 2; do
SL_FUSE_OUTPUT_BWD=1 python bench.py --no-cpu-baseline --no-also --steps 40 > gpurun_out/r03e_on_$r.json
SL_FUSE_OUTPUT_BWD=0 python bench.py --no-cpu-baseline --no-also --steps 40 > gpurun_out/r03e_off_$r.json
done
