"""Data-parallel training of a NeuralCDE across the GPUs of one node: one process per GPU, the batch
sharded by sample (samples never interact inside the solve), weights replicated, and
exactly ONE exchange step per training step -- a sum all-reduce of the flat fp32 gradient over
RCCL/xGMI (``torch.distributed`` backend "nccl"; "gloo" for the CPU tests).

The reference has no counterpart (it fans independent experiments out with GNU parallel,
/root/reference/experiments/runs.py:63-73); this is new work required by BASELINE.json's north_star.
The message is tiny (|theta| = 23,937 floats ~ 96 KB at cfg2/3), i.e. latency-bound: it is sent as one
flat bucket so the collective is a single launch.
"""
import os

import torch
import torch.distributed as dist


def env_world():
    return int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))


def init_process_group(backend=None):
    """Initialise from the torchrun environment (RANK/WORLD_SIZE/MASTER_*); no-op for a single process."""
    rank, local_rank, world = env_world()
    if world > 1 and not dist.is_initialized():
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, local_rank, world


def shard_bounds(n_total, rank, world):
    """Contiguous shard [lo, hi) of n_total samples for `rank` (sizes differ by at most one)."""
    base, rem = divmod(n_total, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


class FlatGradAllReduce:
    """One flat fp32 bucket holding every parameter gradient; ``reduce()`` = one all-reduce(sum) / world.

    With ``average=True`` the result is the gradient of the mean loss over the GLOBAL batch when every
    rank's loss is the mean over its own equally sized shard.
    """

    def __init__(self, params, average=True, single=None):
        self.params = [p for p in params if p.requires_grad]
        self.average = average
        # single process: no bucket aliasing at all (autograd assigns fresh .grad tensors: no zero fill, no accumulate kernels)
        self.single = (not (dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1)) if single is None else single
        n = sum(p.numel() for p in self.params)
        dev = self.params[0].device
        self.flat = torch.zeros(n, dtype=torch.float32, device=dev)
        if not self.single:
            # point every .grad into the bucket so backward writes land there without a pack step
            off = 0
            for p in self.params:
                p.grad = self.flat[off:off + p.numel()].view_as(p)
                off += p.numel()

    def zero(self):
        if self.single:      # one process: nothing to exchange -- let autograd assign fresh .grad tensors (no zero fill,
            for p in self.params:   # no accumulate kernels)
                p.grad = None
        else:
            self.flat.zero_()

    def reduce(self):
        if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
            # autograd may have replaced .grad tensors (it accumulates in place when .grad exists, so this
            # is normally a no-op); re-pack defensively if a grad no longer aliases the bucket
            off = 0
            for p in self.params:
                n = p.numel()
                if p.grad is not None and p.grad.data_ptr() != self.flat[off:off + n].data_ptr():
                    self.flat[off:off + n].copy_(p.grad.reshape(-1))
                    p.grad = self.flat[off:off + n].view_as(p)
                off += n
            dist.all_reduce(self.flat, op=dist.ReduceOp.SUM)
            if self.average:
                self.flat.div_(dist.get_world_size())
        return self.flat

    def gathered(self):
        """The flat gradient as one tensor (the bucket itself in the multi-process case; packed on demand otherwise)."""
        if self.single:
            return torch.cat([(p.grad if p.grad is not None else torch.zeros_like(p)).reshape(-1) for p in self.params])
        return self.flat


def train_step(model, bucket, optimizer, inputs, targets, loss_fn):
    """forward (fused kernel) -> loss -> adjoint backward (fused kernel) -> gradient all-reduce -> optimizer."""
    bucket.zero()
    out = model(inputs)
    loss = loss_fn(out, targets)
    loss.backward()
    bucket.reduce()
    optimizer.step()
    return loss.detach()
