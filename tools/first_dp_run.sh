#!/bin/bash
# tools/first_dp_run.sh -- the first real N > 1 run as ONE command.
#
#   tools/first_dp_run.sh [N=8] [OUT=gpurun_out/first_dp_run]
#
# Runs `bench.py --gpus N` for BASELINE configurations 3 (8 x 32 utterances, config 4's data-parallel step) and 5 (long form)
# over the knobs nobody could measure on one GPU, and prints one table.  Nothing here changes results: every combination ends
# its steps with bit-identical weights on all ranks (bench.py checks it: `identical` column).
#
#   column / knob            what the run decides                                                  DESIGN.md
#   ----------------------   --------------------------------------------------------------------  ---------
#   exchange = allreduce     bucketed sum all-reduce, fused Adam on every rank                      section 5
#            = shard         reduce-scatter + Adam on the rank's slice + all-gather (1/N of the
#                            optimizer's 830 MB per rank, same bytes per link) -- pays if Adam's
#                            0.14 ms is worth more than the second collective's launch latency
#   comm_cus = 0 / 32        sl_set_available_cus(256 - n) for backward's grid choosers while a     section 5 table
#                            bucket is on the wire: slower under the one-GPU stand-in (+0.14 ms);
#                            RCCL's real channel kernels may own fewer CUs for less time
#   split    = 0 / 1         Engine.split_last_bucket: inner_conv_4..7 close early (7 of the 19 MB  section 5c
#                            nothing covers) for +2 launches per step -- estimated +-50 us
#   rccl     = default /     NCCL_ALGO / channel counts: a ring is per-link bound on point-to-point section 5
#              tree / ...    xGMI (7 links x ~76 GB/s per direction); direct / tree algorithms and
#                            more channels use all links at the price of more CUs
#
# Needs N GPUs on this node.  SL_BENCH_SHARE_GPU=1 runs the same matrix with N ranks sharing ONE GPU over gloo (control
# flow only, never a measurement; tests/test_gpu_round6.py::test_first_dp_run_matrix_control_flow does that with N = 8).
set -u
N=${1:-8}
OUT=${2:-gpurun_out/first_dp_run}
STEPS=${STEPS:-50}
WARMUP=${WARMUP:-10}
CONFIGS=${CONFIGS:-"3 5"}
cd "$(dirname "$0")/.."
mkdir -p "$OUT"
export HSA_ENABLE_IPC_MODE_LEGACY=0 MASTER_ADDR=127.0.0.1

# name | environment
RCCL_VARIANTS=${RCCL_VARIANTS:-"default| ring|NCCL_ALGO=Ring tree|NCCL_ALGO=Tree ch32|NCCL_MIN_NCHANNELS=32,NCCL_MAX_NCHANNELS=32 ch8|NCCL_MIN_NCHANNELS=8,NCCL_MAX_NCHANNELS=8"}

run_one() {  # config exchange comm_cus split rccl_name rccl_env
    local cfg=$1 exch=$2 cus=$3 split=$4 rname=$5 renv=$6
    local tag="c${cfg}_${exch}_cus${cus}_split${split}_${rname}"
    local args=(--gpus "$N" --config "$cfg" --steps "$STEPS" --warmup "$WARMUP" --no-cpu-baseline --no-also --profile-steps 1
                --comm-cus "$cus")
    [ "$exch" = shard ] && args+=(--shard-optimizer)
    [ "$split" = 1 ] && args+=(--split-last-bucket)
    local envs=()
    IFS=',' read -ra kv <<< "$renv"
    for e in "${kv[@]}"; do [ -n "$e" ] && envs+=("$e"); done
    env "${envs[@]}" python bench.py "${args[@]}" > "$OUT/$tag.json" 2> "$OUT/$tag.err"
    echo "$? $tag" >> "$OUT/status.txt"
}

: > "$OUT/status.txt"
for cfg in $CONFIGS; do
    # the single-GPU rate of the same configuration: the denominator of the efficiency column
    python bench.py --gpus 1 --config "$cfg" --steps "$STEPS" --warmup "$WARMUP" --no-cpu-baseline --no-also --profile-steps 1 \
        > "$OUT/c${cfg}_single.json" 2> "$OUT/c${cfg}_single.err"
    for exch in ${EXCHANGES:-allreduce shard}; do
        for cus in ${COMM_CUS:-0 32}; do
            for split in ${SPLITS:-0 1}; do
                run_one "$cfg" "$exch" "$cus" "$split" default ""
            done
        done
    done
    # the RCCL environment on the default combination only (it is orthogonal to the engine's knobs)
    for v in $RCCL_VARIANTS; do
        name=${v%%|*}; renv=${v#*|}
        [ "$name" = default ] && continue
        run_one "$cfg" allreduce 0 0 "$name" "$renv"
    done
done
python tools/first_dp_table.py "$OUT" "$N" | tee "$OUT/table.txt"
