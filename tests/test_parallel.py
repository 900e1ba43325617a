"""CPU tests (gloo, world_size 2) of the data-parallel gradient exchange in speechless_amd/parallel.py.

Each rank computes, with the CPU oracle (allowed in tests), the gradient of its utterance shard scaled by
1/(B_local * world) -- exactly what Engine.train_step_resident asks the CTC kernel for -- packs it into a flat fp32
buffer, and the bucketed all-reduce must reproduce the single-process gradient of the global batch."""
import os
import socket
import sys
from pathlib import Path

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = Path(__file__).resolve().parent.parent
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))


def _toy_problem():
    from oracle import w2l_oracle as o
    specs = o.layer_specs(4, 5, main_filter_count=6, out_filter_count=8, striding_kernel=6, inner_kernel=3,
                          big_kernel=4, inner_count=2)
    weights = o.glorot_uniform_weights(specs, 2, np.float64)
    x = np.random.RandomState(0).randn(4, 20, 4)
    labels = np.array([[0, 1, 2], [3, 3, -1], [1, -1, -1], [2, 0, 2]])
    return specs, weights, x, labels, [10, 9, 8, 10], [3, 2, 1, 3]


def _flat(grads):
    return np.concatenate([np.concatenate([dw.ravel(), db.ravel()]) for dw, db in grads]).astype(np.float32)


def _worker(rank, world, port, out_dir, compress=None, shard=False):
    from oracle import w2l_oracle as o
    from speechless_amd.parallel import GradBucketReducer, shard_range
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    specs, weights, x, labels, pred_len, lab_len = _toy_problem()
    lo, hi = shard_range(x.shape[0], rank, world)
    r = o.loss_and_gradients(specs, weights, x[lo:hi], labels[lo:hi], pred_len[lo:hi], lab_len[lo:hi])
    # the oracle scales by 1/B_local (mean over the shard); the engine uses 1/(B_local*world)
    flat = torch.from_numpy(_flat(r["grads"]) / world)
    n = flat.numel()
    ranges = [(n // 3, n), (0, n // 3)]  # "late layers first" bucket order, like Engine.bucket_ranges()
    if shard:  # bucket lengths must divide by the world size: trim the flat vector to a multiple of 2 * world
        n = n // (3 * 2 * world) * (3 * 2 * world)
        flat = flat[:n].clone()
        ranges = [(n // 3, n), (0, n // 3)]
    reducer = GradBucketReducer(flat, ranges, compress=compress, shard_optimizer=shard)
    assert reducer.world_size == world
    reducer.reduce_bucket(0)
    reducer.reduce_bucket(1)
    reducer.wait_all()
    if shard:
        # each rank now holds the SUM on its own slice of every bucket only; a stand-in "optimizer" (x -> 2x) on that
        # slice, then the all-gather: every rank must end up with 2 * (all-reduced gradient) everywhere
        for lo, hi in ranges:
            slo, shi = reducer.shard_of(lo, hi)
            flat[slo:shi] *= 2.0
        reducer.gather_bucket(0, flat)
        reducer.gather_bucket(1, flat)
        reducer.wait_all()
    np.save(os.path.join(out_dir, "rank{}.npy".format(rank)), flat.numpy())
    dist.destroy_process_group()


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def test_two_rank_allreduce_equals_single_process_gradient(tmp_path):
    from oracle import w2l_oracle as o
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    specs, weights, x, labels, pred_len, lab_len = _toy_problem()
    ref = _flat(o.loss_and_gradients(specs, weights, x, labels, pred_len, lab_len)["grads"])
    for rank in range(world):
        got = np.load(str(tmp_path / "rank{}.npy".format(rank)))
        np.testing.assert_allclose(got, ref, rtol=2e-5, atol=1e-7)


def test_bf16_compressed_allreduce_within_bf16_tolerance(tmp_path):
    """compress="bf16" (optional, off by default): the reduced gradient is the exact one up to bf16 rounding of each
    rank's contribution and of the sum (2^-9 relative per rounding; measured 4e-3 rel-L2) and identical on both ranks."""
    from oracle import w2l_oracle as o
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path), "bf16"), nprocs=world, join=True)
    specs, weights, x, labels, pred_len, lab_len = _toy_problem()
    ref = _flat(o.loss_and_gradients(specs, weights, x, labels, pred_len, lab_len)["grads"])
    got = [np.load(str(tmp_path / "rank{}.npy".format(rank))) for rank in range(world)]
    assert np.array_equal(got[0], got[1])
    assert np.linalg.norm(got[0] - ref) / np.linalg.norm(ref) < 1e-2
    assert np.linalg.norm(got[0] - ref) > 0  # it did go through bf16


def test_sharded_exchange_equals_allreduce(tmp_path):
    """shard_optimizer: reduce-scatter + (slice-local update) + all-gather of the same buffer reproduces what an
    all-reduce followed by the update on every rank gives, on both ranks."""
    from oracle import w2l_oracle as o
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path), None, True), nprocs=world, join=True)
    specs, weights, x, labels, pred_len, lab_len = _toy_problem()
    ref = _flat(o.loss_and_gradients(specs, weights, x, labels, pred_len, lab_len)["grads"])
    got = [np.load(str(tmp_path / "rank{}.npy".format(rank))) for rank in range(world)]
    assert np.array_equal(got[0], got[1])
    np.testing.assert_allclose(got[0], 2.0 * ref[:got[0].size], rtol=2e-5, atol=1e-7)
    from speechless_amd.parallel import GradBucketReducer
    with pytest.raises(ValueError):
        GradBucketReducer(torch.zeros(8), [(0, 8)], shard_optimizer=True, compress="bf16")


def test_shard_range_partitions_everything():
    from speechless_amd.parallel import shard_range
    for total, world in [(256, 8), (64, 8), (10, 4), (3, 2), (1, 1)]:
        covered = []
        for r in range(world):
            lo, hi = shard_range(total, r, world)
            covered += list(range(lo, hi))
        assert covered == list(range(total))
    assert shard_range(256, 3, 8) == (96, 128)  # rank r takes utterances [32r, 32r+32) (config 4)


def test_single_process_reducer_is_a_no_op():
    from speechless_amd.parallel import GradBucketReducer
    flat = torch.arange(10, dtype=torch.float32)
    reducer = GradBucketReducer(flat, [(5, 10), (0, 5)])
    reducer.reduce_bucket(0)
    reducer.reduce_bucket(1)
    reducer.wait_all()
    assert torch.equal(flat, torch.arange(10, dtype=torch.float32))


# ------------------------------------------------------------------------------------------ the real engine, two ranks
def _engine_case():
    from oracle import w2l_oracle as o
    from speechless_amd.engine import wav2letter_layer_specs
    specs = wav2letter_layer_specs(128, 29)
    weights = o.glorot_uniform_weights(o.layer_specs(128, 29), seed=2, dtype=np.float32)
    rng = np.random.RandomState(7)
    x = rng.randn(4, 120, 128).astype(np.float32)
    lab_len = np.array([9, 4, 12, 7], dtype=np.int32)
    labels = -np.ones((4, 12), dtype=np.int32)
    for i, n in enumerate(lab_len):
        labels[i, :n] = rng.randint(0, 28, size=n)
    pred_len = np.array([60, 57, 60, 55], dtype=np.int32)
    return specs, weights, x, labels, lab_len, pred_len


def _engine_worker(rank, world, port, out_dir, shard=False, dtype="f32"):
    from speechless_amd.engine import Engine
    from speechless_amd.parallel import GradBucketReducer, shard_range
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)  # both ranks share cuda:0: gloo moves the bytes
    specs, weights, x, labels, lab_len, pred_len = _engine_case()
    eng = Engine(specs, 29, dtype=dtype, device="cuda:0", lr=1e-3)
    eng.set_weights(weights)
    ranges = eng.bucket_ranges()
    assert len(ranges) == 4 and sum(hi - lo for lo, hi in ranges) == eng.param_numel  # (fp32 / bf16x3 paths: 4 buckets) cover every parameter
    reducer = GradBucketReducer(eng.grads, ranges, shard_optimizer=shard)
    lo, hi = shard_range(x.shape[0], rank, world)
    for _ in range(2):
        eng.train_step(x[lo:hi], labels[lo:hi], lab_len[lo:hi], pred_len[lo:hi], reducer)
    torch.cuda.synchronize()
    np.savez(os.path.join(out_dir, "rank{}.npz".format(rank)), *[w for w, _ in eng.get_weights()])
    # the optimizer state as Wav2Letter.save_optimizer_state would write it: with the sharded optimizer a collective that
    # gathers every rank's slices of the moments
    state = eng.get_optimizer_state()
    np.savez(os.path.join(out_dir, "opt{}.npz".format(rank)), *([m for m, _ in state["m"]] + [v for v, _ in state["v"]]))
    dist.destroy_process_group()


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", ["f32", "bf16x3"])
@pytest.mark.parametrize("shard", [False, True])
def test_two_engine_ranks_equal_one_rank_on_the_global_batch(tmp_path, shard, dtype):
    """Two processes, each running the real HIP engine on its utterance shard with the bucketed, overlapped gradient
    all-reduce (gloo transport, both on the one GPU of the test box), must end up with the weights of a single process
    that trained on the global batch: same gradient scale (1 / (B_local * world)), same bucket ranges, same Adam -- on
    the exact-fp32 path and (round 4) on the bf16x3 path, the fast path that meets the gradient bar.  With the sharded
    optimizer the optimizer state each rank would save (get_optimizer_state: a collective then) is the full state, equal
    on both ranks and to the single process's."""
    from speechless_amd.engine import Engine
    world = 2
    mp.spawn(_engine_worker, args=(world, _free_port(), str(tmp_path), shard, dtype), nprocs=world, join=True)
    specs, weights, x, labels, lab_len, pred_len = _engine_case()
    eng = Engine(specs, 29, dtype=dtype, device="cuda:0", lr=1e-3)
    eng.set_weights(weights)
    for _ in range(2):
        eng.train_step(x, labels, lab_len, pred_len)
    torch.cuda.synchronize()
    ref = [w for w, _ in eng.get_weights()]
    moved = 0.0
    ranks = [np.load(str(tmp_path / "rank{}.npz".format(rank))) for rank in range(world)]
    for i in range(len(ref)):  # the ranks agree bit for bit (same reduced gradient / the same gathered masters)
        assert np.array_equal(ranks[0]["arr_{}".format(i)], ranks[1]["arr_{}".format(i)])
    for rank in range(world):
        got = ranks[rank]
        for i, r in enumerate(ref):
            g = got["arr_{}".format(i)]
            # identical up to the fp32 summation order of the gradient (two partial sums added by the all-reduce); Adam's
            # m / sqrt(v) amplifies that for the few elements whose gradient is ~0, hence a bound relative to the step
            step_size = np.abs(r - weights[i][0]).max()
            diff = np.abs(g - r)
            assert diff.max() <= 2e-2 * step_size and np.mean(diff > 1e-3 * step_size) < 1e-3, (rank, i, diff.max())
            moved = max(moved, float(np.abs(r - weights[i][0]).max()))
    assert moved > 1e-4  # the steps did change the weights
    state = eng.get_optimizer_state()
    want = [m for m, _ in state["m"]] + [v for v, _ in state["v"]]
    opts = [np.load(str(tmp_path / "opt{}.npz".format(rank))) for rank in range(world)]
    for i, w in enumerate(want):
        a, b = opts[0]["arr_{}".format(i)], opts[1]["arr_{}".format(i)]
        assert np.array_equal(a, b), i                                # every rank holds (and would save) the same state
        assert np.linalg.norm(a.astype(np.float64) - w) <= 1e-4 * np.linalg.norm(w) + 1e-30, i   # ... the FULL state


@pytest.mark.gpu
@pytest.mark.parametrize("shard", [False, True])
def test_bench_two_ranks_emit_the_data_parallel_diagnostics(shard):
    """bench.py's multi-rank control flow and its `data_parallel` object (world size, bucket sizes, all-reduce alone,
    exposed communication, identical reduced gradients / weights on all ranks), exercised with two ranks sharing the
    one GPU of the test box over gloo (SL_BENCH_SHARE_GPU=1: a test hook, never a measurement)."""
    import json
    import subprocess
    import sys
    from pathlib import Path
    root = Path(__file__).resolve().parent.parent
    env = dict(os.environ, SL_BENCH_SHARE_GPU="1", MASTER_ADDR="127.0.0.1")
    # plain `python bench.py --gpus 2`: the script starts its own ranks (the shape of the driver's command)
    cmd = [sys.executable, str(root / "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--profile-steps", "1"] + \
        (["--shard-optimizer"] if shard else [])
    res = subprocess.run(cmd, capture_output=True, text=True, env=env, cwd=str(root), timeout=900)
    assert res.returncode == 0, res.stderr[-2000:]
    line = [l for l in res.stdout.splitlines() if l.startswith("{")][-1]
    out = json.loads(line)
    assert out["n_gpus"] == 2 and out["config"]["global_batch"] == 64 and out["scaling"] == "weak"
    dp = out["data_parallel"]
    assert dp["world_size"] == 2 and dp["reduced_gradients_and_weights_identical_on_all_ranks"] is True
    assert len(dp["bucket_bytes"]) == 3 and sum(dp["bucket_bytes"]) > 90e6 and dp["sharded_optimizer"] is shard
    assert dp["allreduce_alone_ms"] > 0 and dp["step_ms_with_allreduce"] > 0 and np.isfinite(dp["gradient_checksum"])
    assert len(line) < 4000 and len(json.dumps(dp)) < 2000  # the driver extracts the LAST stdout line: it must stay small
    full = json.loads((root / out["detail"]).read_text())["data_parallel"]  # everything else: the detail file
    assert full["bucket_layers"][0] == ["big_conv_2", "output_conv"] and full["bucket_layers"][-1][0] == "striding_conv" \
        and len(full["bucket_layers"][-1]) == 8  # (one launch writes striding_conv + inner_conv_1..7: one bucket)



@pytest.mark.gpu
@pytest.mark.parametrize("config", [3, 5])
def test_bench_eight_ranks_control_flow(config):
    """`bench.py --gpus 8` has never run on 8 GPUs; its control flow has -- eight ranks sharing the test
    box's one GPU over gloo (SL_BENCH_SHARE_GPU=1, never a measurement), configurations 3 and 5: the last stdout line parses,
    is small, names 8 GPUs and the global batch, every rank ended the step with identical weights, and for configuration 5
    the steps came from ONE global corpus dealt across the ranks (batching.steps_for_ranks)."""
    import json
    import subprocess
    import sys
    from pathlib import Path
    root = Path(__file__).resolve().parent.parent
    env = dict(os.environ, SL_BENCH_SHARE_GPU="1", MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, str(root / "bench.py"), "--gpus", "8", "--steps", "2", "--warmup", "1", "--profile-steps", "1",
           "--config", str(config)]
    res = subprocess.run(cmd, capture_output=True, text=True, env=env, cwd=str(root), timeout=1500)
    assert res.returncode == 0, res.stderr[-2000:]
    line = [l for l in res.stdout.splitlines() if l.startswith("{")][-1]
    assert len(line) < 4000
    out = json.loads(line)
    per_gpu = 32 if config == 3 else 8
    assert out["n_gpus"] == 8 and out["config"]["global_batch"] == 8 * per_gpu and out["scaling"] == "weak"
    assert out["steps"] == 2 and out["warmup"] == 1 and out["value"] > 0
    dp = out["data_parallel"]
    assert dp["world_size"] == 8 and dp["reduced_gradients_and_weights_identical_on_all_ranks"] is True
    assert len(json.dumps(dp)) < 2000
    if config == 5:
        assert out["config"]["step_formation"].startswith("one global corpus")
