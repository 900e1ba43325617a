# usage (on the GPU box, via gpurun): bash tools/profile_round.sh <tag>      e.g. r01f
# rocprofv3 kernel statistics of the default bench.py run + the bench line itself + PMC traffic of the roofline kernel;
# everything lands under gpurun_out/<tag>_* (copy what should be judged into profiles/).
TAG=$1
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 600 python bench.py > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${TAG}_prof -o p -- python bench.py --no-cpu-baseline > gpurun_out/${TAG}_bench_under_rocprof.json 2> gpurun_out/${TAG}_prof.err
cp gpurun_out/${TAG}_prof/*kernel_stats.csv gpurun_out/${TAG}_kernel_stats.csv 2>/dev/null
bash tools/pmc_traffic.sh > gpurun_out/${TAG}_pmc_traffic.log 2>&1
cp gpurun_out/pmc_traffic.json gpurun_out/${TAG}_pmc_traffic_wgrad_ilv.json
tail -1 gpurun_out/${TAG}_bench.json | cut -c1-250
head -8 gpurun_out/${TAG}_kernel_stats.csv | cut -c1-160
