"""Data-parallel gradient exchange: one process per GPU, utterances sharded across ranks, ONE exchange step per
optimisation step -- a sum all-reduce of the flat fp32 gradient buffer (RCCL over xGMI when the tensors live in HBM;
`torch.distributed` backend "nccl" IS RCCL on ROCm).

The reference (juliuskunze/speechless) has no distributed code at all (single TF session, main.py:14-24); this is new.
Utterances are independent through forward, CTC and backward and the loss is a mean over the batch (net.py:389), so
gradients add: rank r computes d(sum_b loss_b)/dW scaled by 1/(B_local * world) and the all-reduce sums the ranks.

xGMI is point-to-point, so one big ring all-reduce is per-link bound: the exchange is split into the buckets of
Engine.bucket_ranges() and bucket 0 (output layers, 81 % of the bytes, ready first in backward) is reduced on a side
stream while the remaining layers are still in backward.

Device-agnostic on purpose: the same class runs over gloo on CPU tensors in tests/test_parallel.py.
"""
import torch
import torch.distributed as dist


class GradBucketReducer:
    def __init__(self, flat_grads, ranges, process_group=None, overlap=True, force=False, compress=None,
                 shard_optimizer=False, comm_cus=None):
        """force: issue the collectives even in a world of one rank (tests exercise the stream / event choreography and
        the RCCL call on a single GPU that way).
        compress: None (fp32 on the wire: the reduced gradient is the exact sum, identical on every rank) or "bf16"
        (each rank's bucket is rounded to bf16, summed in bf16 by the collective and widened again: half the bytes per
        link, ~3 significant digits per gradient element -- an option for link-bound scaling, off by default; the
        result is still identical on every rank).
        shard_optimizer: reduce-SCATTER each bucket instead of all-reducing it -- rank r ends up with the summed gradient
        of slice r of every bucket only (shard_of), the engine runs Adam on that slice (1/world of the optimizer's HBM
        traffic per rank) and gather_bucket() all-gathers the updated fp32 masters in place.  Same bytes on the wire as the
        all-reduce (a ring all-reduce IS this reduce-scatter followed by this all-gather), the optimizer between the two
        halves.  Needs every bucket length to be a multiple of the world size."""
        if compress not in (None, "bf16"):
            raise ValueError("compress must be None or 'bf16'")
        # CUs the collectives are expected to own while a bucket is on the wire (one work-group per RCCL channel): the engine
        # then sizes the grids of backward's MFMA kernels for the rest (Engine.comm_cus, sl_set_available_cus).  Default:
        # SL_COMM_CUS, else 0 = no hint -- under a stand-in that owns 32 / 64 CUs the re-planned grids were measured SLOWER
        # than the whole-chip ones (2.57 / 2.53 against 2.39 / 2.32 ms per step, profiles/r04_comm_interference_exclusive*.json:
        # the exchange owns its CUs for a third of backward, the hint costs all of it).  Kept as a knob for the first N > 1 run.
        import os
        if comm_cus is None:
            comm_cus = int(os.environ.get("SL_COMM_CUS", "0"))
        self.comm_cus = int(comm_cus)
        if shard_optimizer and compress:
            raise ValueError("shard_optimizer moves fp32 slices; it does not combine with compress")
        self.compress = compress
        self.force = force
        self.flat = flat_grads
        self.ranges = list(ranges)
        self.group = process_group
        self.world_size = dist.get_world_size(process_group) if dist.is_initialized() else 1
        self.cuda = flat_grads.is_cuda
        self.overlap = overlap and self.cuda
        self.comm_stream = torch.cuda.Stream(device=flat_grads.device) if self.overlap else None
        self._pending = []
        # measurement only (bench.py's exposed-communication figure): keep the stream / event choreography of a step
        # but leave the collective itself out
        self.skip_collective = False
        self.shard_optimizer = bool(shard_optimizer)
        self.rank = dist.get_rank(process_group) if dist.is_initialized() else 0
        if self.shard_optimizer:
            for lo, hi in self.ranges:
                if (hi - lo) % self.world_size:
                    raise ValueError("shard_optimizer: bucket [{}, {}) is not a multiple of the world size {}".format(
                        lo, hi, self.world_size))

    def shard_of(self, lo, hi):
        """The slice of [lo, hi) this rank owns under shard_optimizer."""
        per = (hi - lo) // self.world_size
        return lo + self.rank * per, lo + (self.rank + 1) * per

    def reduce_bucket(self, index):
        """Called when every kernel writing bucket `index` has been enqueued on the current stream."""
        if self.world_size == 1 and not self.force:
            return
        lo, hi = self.ranges[index]
        view = self.flat[lo:hi]
        self._exchange(view, self._reduce_scatter if self.shard_optimizer else None)

    def gather_bucket(self, index, flat_params):
        """shard_optimizer: all-gathers bucket `index` of flat_params (same layout as the gradient buffer) in place, every
        rank contributing its own slice, behind everything enqueued on the current stream so far.  The completion is
        queued behind the outstanding buckets (wait_next / wait_all)."""
        if self.world_size == 1 and not self.force:
            return
        lo, hi = self.ranges[index]
        self._exchange(flat_params[lo:hi], self._all_gather)

    def _reduce_scatter(self, view):
        per = view.numel() // self.world_size
        dist.reduce_scatter_tensor(view[self.rank * per:(self.rank + 1) * per], view, op=dist.ReduceOp.SUM,
                                   group=self.group)

    def _all_gather(self, view):
        per = view.numel() // self.world_size
        dist.all_gather_into_tensor(view, view[self.rank * per:(self.rank + 1) * per], group=self.group)

    def _exchange(self, view, collective):
        """runs collective(view) (default: the sum all-reduce) on the communication stream behind the current stream"""
        if self.overlap:
            ready = torch.cuda.Event()
            ready.record(torch.cuda.current_stream(self.flat.device))
            with torch.cuda.stream(self.comm_stream):
                self.comm_stream.wait_event(ready)
                if not self.skip_collective:
                    (collective or self._all_reduce)(view)
                done = torch.cuda.Event()
                done.record(self.comm_stream)
            self._pending.append(done)
        elif not self.skip_collective:
            if collective is not None:
                collective(view)
            elif self.compress:
                self._all_reduce(view)
            else:
                work = dist.all_reduce(view, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
                self._pending.append(work)

    def _all_reduce(self, view):
        if self.compress == "bf16":
            wire = view.to(torch.bfloat16)
            dist.all_reduce(wire, op=dist.ReduceOp.SUM, group=self.group)
            view.copy_(wire)
        else:
            dist.all_reduce(view, op=dist.ReduceOp.SUM, group=self.group)

    def wait_next(self):
        """Makes the current stream (or the host, on CPU) wait for the OLDEST outstanding bucket only."""
        if not self._pending:
            return
        p = self._pending.pop(0)
        if self.overlap:
            torch.cuda.current_stream(self.flat.device).wait_event(p)
        else:
            p.wait()

    def wait_all(self):
        """Makes the current stream (or the host, on CPU) wait for every outstanding bucket."""
        for p in self._pending:
            if self.overlap:
                torch.cuda.current_stream(self.flat.device).wait_event(p)
            else:
                p.wait()
        self._pending = []


def shard_range(total, rank, world_size):
    """Contiguous utterance shard of rank `rank` (rank r takes utterances [r*n/world, (r+1)*n/world))."""
    per = total // world_size
    rem = total % world_size
    lo = rank * per + min(rank, rem)
    return lo, lo + per + (1 if rank < rem else 0)
