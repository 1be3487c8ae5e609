"""CPU: the data-parallel path with world_size 2 over gloo — bucketed flat-gradient
average, batch sharding, and equality of the sharded gradient with the single-process one
(computed with the oracle; the HIP path is not needed for the exchange logic)."""
import os
import sys

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from helpers import O, ROOT, pkg


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(2)
    P = pkg("parallel")
    Fl = pkg("utils.flat")
    G = pkg("esrgan.generator")
    # model replica with deterministic weights; flat parameter storage
    g = G.GeneratorESRGAN(num_features=16, num_blocks=1)
    O.fill_state_dict_(g.state_dict(), salt=21)
    flat = Fl.flatten_parameters(g)
    assert flat.numel() == sum(p.numel() for p in g.parameters())
    assert next(g.parameters()).data_ptr() == flat.data_ptr()
    sd = {k: v for k, v in g.state_dict().items()}
    lr = O.hash_input("ddp.lr", (4, 3, 8, 8))
    hr = O.hash_input("ddp.hr", (4, 3, 32, 32))
    lr_s, hr_s = P.shard_batch(lr, rank, world), P.shard_batch(hr, rank, world)
    params = [p for p in g.parameters()]
    # local gradient of the local mean loss (oracle arithmetic), flattened in parameter order
    out = O.generator_forward(lr_s, {k: p for (k, _), p in zip(g.named_parameters(), params)})
    loss = O.l1_loss(out, hr_s)
    grads = torch.autograd.grad(loss, params)
    fg = torch.cat([x.reshape(-1) for x in grads])
    sync = P.FlatGradAllReduce(world, bucket_mb=0.05)        # several buckets
    assert len(sync.buckets(fg.numel())) > 3
    sync(fg)
    # the overlapped form: ranges handed over in backward order (last parameters first), merged
    # into buckets, averaged — the same result as the one-shot form
    fg2 = torch.cat([x.reshape(-1) for x in grads])
    ov = P.OverlappedGradAllReduce(world, bucket_mb=0.05)
    ov.begin(fg2)
    cuts = list(range(0, fg2.numel(), 4001)) + [fg2.numel()]
    for lo, hi in reversed(list(zip(cuts[:-1], cuts[1:]))):
        ov.ready(lo, hi)
    ov.finish()
    assert torch.equal(fg2, fg)
    # replicas that start apart are made equal to rank 0's
    t = torch.full((1000,), float(rank + 1))
    u = torch.arange(5, dtype=torch.float32) * (rank + 1)
    P.broadcast_from_rank0([t, u])
    assert torch.equal(t, torch.full((1000,), 1.0)) and torch.equal(u, torch.arange(5, dtype=torch.float32))
    P.barrier()
    if rank == 0:
        out_full = O.generator_forward(lr, {k: p for (k, _), p in zip(g.named_parameters(), params)})
        gfull = torch.autograd.grad(O.l1_loss(out_full, hr), params)
        ref = torch.cat([x.reshape(-1) for x in gfull])
        q.put(float((fg - ref).abs().max() / ref.abs().max()))
    dist.barrier()
    dist.destroy_process_group()


def test_flat_grad_allreduce_matches_single_process():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(300)
        assert p.exitcode == 0
    err = q.get(timeout=10)
    assert err < 1e-5, err


def test_shard_batch_rejects_uneven():
    import pytest
    P = pkg("parallel")
    with pytest.raises(ValueError):
        P.shard_batch(torch.zeros(5, 3, 4, 4), 0, 2)
    assert P.shard_batch(torch.arange(8).reshape(8, 1), 1, 2).flatten().tolist() == [4, 5, 6, 7]
