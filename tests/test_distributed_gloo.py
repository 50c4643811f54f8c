"""N>1 path on CPU: world_size 2 over gloo.  The solve itself needs a GPU (no CPU fallback), so the
data-parallel plumbing -- sharding, the single flat-bucket all-reduce, identical replicas after the
optimizer step -- is exercised with a small torch stand-in model; the same code runs under RCCL."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import ncde_amd
from ncde_amd import distributed as D


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _make_model():
    torch.manual_seed(0)
    return torch.nn.Sequential(torch.nn.Linear(6, 5), torch.nn.Tanh(), torch.nn.Linear(5, 1))


def _data(lo, hi):
    x = ncde_amd.data.normal(3, 16 * 6, stream=1).reshape(16, 6).astype(np.float32)
    y = (ncde_amd.data.uniform01(3, 16, stream=2) > 0.5).astype(np.float32).reshape(16, 1)
    return torch.from_numpy(x[lo:hi]), torch.from_numpy(y[lo:hi])


def _worker(rank, world, port, out_dir):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    torch.set_num_threads(1)
    r, _, w = D.init_process_group("gloo")
    assert (r, w) == (rank, world)
    lo, hi = D.shard_bounds(16, rank, world)
    x, y = _data(lo, hi)
    model = _make_model()
    bucket = D.FlatGradAllReduce(model.parameters())
    opt = torch.optim.Adam(model.parameters(), lr=1e-2)
    loss_fn = torch.nn.BCEWithLogitsLoss()
    for _ in range(3):
        D.train_step(model, bucket, opt, x, y, loss_fn)
    flat = torch.cat([p.detach().reshape(-1) for p in model.parameters()])
    torch.save({"params": flat, "grad": bucket.gathered().clone()}, os.path.join(out_dir, "rank%d.pt" % rank))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_data_parallel_matches_single_process(tmp_path):
    port = _free_port()
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    r0 = torch.load(tmp_path / "rank0.pt")
    r1 = torch.load(tmp_path / "rank1.pt")
    assert torch.equal(r0["params"], r1["params"])           # replicas stay identical
    assert torch.equal(r0["grad"], r1["grad"])
    # single process on the full batch (mean loss) == average of the two equal shards' mean-loss gradients
    model = _make_model()
    x, y = _data(0, 16)
    opt = torch.optim.Adam(model.parameters(), lr=1e-2)
    bucket = D.FlatGradAllReduce(model.parameters())
    for _ in range(3):
        D.train_step(model, bucket, opt, x, y, torch.nn.BCEWithLogitsLoss())
    flat = torch.cat([p.detach().reshape(-1) for p in model.parameters()])
    assert torch.allclose(flat, r0["params"], rtol=1e-5, atol=1e-6)
    assert torch.allclose(bucket.gathered(), r0["grad"], rtol=1e-4, atol=1e-6)


def test_shard_bounds_tile_the_batch():
    for n in (1, 7, 16, 4096, 4099):
        for world in (1, 2, 3, 8):
            spans = [D.shard_bounds(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            assert max(hi - lo for lo, hi in spans) - min(hi - lo for lo, hi in spans) <= 1


def test_bucket_aliases_parameter_grads():
    model = _make_model()
    bucket = D.FlatGradAllReduce(model.parameters(), single=False)      # the multi-process layout, exercised in one process
    x, y = _data(0, 16)
    torch.nn.functional.mse_loss(model(x), y).backward()
    off = 0
    for p in model.parameters():
        assert p.grad.data_ptr() == bucket.flat[off:off + p.numel()].data_ptr()   # backward wrote straight into the bucket
        off += p.numel()
    assert float(bucket.flat.abs().sum()) > 0
    bucket.zero()
    assert all(float(p.grad.abs().sum()) == 0 for p in model.parameters())
