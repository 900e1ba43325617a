# usage (on the GPU box, via gpurun): bash tools/profile_round.sh <tag>      e.g. r03
# rocprofv3 kernel statistics of the default bench.py run + the bench line itself + PMC traffic of the roofline kernel +
# the other configurations / end-to-end figures of a round; everything lands under gpurun_out/<tag>_* (copy what should
# be judged into profiles/).
TAG=$1
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python bench.py > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err
cp gpurun_out/bench_detail.json gpurun_out/${TAG}_bench_detail.json
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${TAG}_prof -o p -- python bench.py --no-cpu-baseline --no-also > gpurun_out/${TAG}_bench_under_rocprof.json 2> gpurun_out/${TAG}_prof.err
cp gpurun_out/${TAG}_prof/*kernel_stats.csv gpurun_out/${TAG}_kernel_stats.csv 2>/dev/null
TAG=$TAG bash tools/pmc_traffic.sh > gpurun_out/${TAG}_pmc_traffic.log 2>&1
cp gpurun_out/pmc_traffic.json gpurun_out/${TAG}_pmc_traffic_wgrad_ilv.json
timeout 600 python bench.py --config 2 > gpurun_out/${TAG}_bench_config2.json 2>> gpurun_out/${TAG}_bench.err
timeout 600 python bench.py --config 5 > gpurun_out/${TAG}_bench_config5.json 2>> gpurun_out/${TAG}_bench.err
( timeout 900 python tools/e2e_train_throughput.py; timeout 900 python tools/e2e_train_throughput.py --from-audio --steps 1000 ) 2>&1 | grep -v amdgpu > gpurun_out/${TAG}_e2e.txt
timeout 600 python tools/step_time_by_dtype.py 2>&1 | grep -v amdgpu > gpurun_out/${TAG}_step_time_by_dtype.txt
timeout 600 python tools/split_by_bucket.py --rule 2>&1 | grep -v amdgpu > gpurun_out/${TAG}_split_by_bucket_rule.txt
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${TAG}_prof5 -o p -- python bench.py --config 5 --no-cpu-baseline > gpurun_out/${TAG}_bench_config5_under_rocprof.json 2>> gpurun_out/${TAG}_prof.err
cp gpurun_out/${TAG}_prof5/*kernel_stats.csv gpurun_out/${TAG}_kernel_stats_config5.csv 2>/dev/null
rm -rf gpurun_out/${TAG}_prof gpurun_out/${TAG}_prof5 gpurun_out/pmc_traffic
tail -1 gpurun_out/${TAG}_bench.json | cut -c1-250
head -12 gpurun_out/${TAG}_kernel_stats.csv | cut -c1-160
