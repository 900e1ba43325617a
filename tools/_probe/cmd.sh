cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export SL_BENCH_SHARE_GPU=1
for extra in "" "--shard-optimizer"; do
timeout 900 python bench.py --gpus 8 --steps 3 --warmup 1 $extra > gpurun_out/dp8.json 2> gpurun_out/dp8.err; echo "rc=$?"
python - <<'PY'
import json
try:
    d=json.loads(open('gpurun_out/dp8.json').read().strip().splitlines()[-1])
    dp=d['data_parallel']
    print(d['n_gpus'], d['config']['global_batch'], round(d['value'],1), dp['world_size'], dp['backend'], dp['sharded_optimizer'], dp['reduced_gradients_and_weights_identical_on_all_ranks'], [round(x/1e6,1) for x in dp['bucket_bytes']], dp['bucket_layers'][-1][:2])
except Exception as e:
    print("ERR", e); print(open('gpurun_out/dp8.err').read()[-1500:])
PY
done
