"""World-size-2 check of the multi-GPU decomposition on CPU (gloo).

Each rank renders its sample range (raytracing_amd.shard_samples, the rule bench.py uses) into an f32 sum image and
the images are sum-reduced to rank 0, exactly the data flow of the RCCL path.  The renderer here is the CPU oracle,
which keys its random streams on the global sample index the same way the kernels do, so the reduced image must
equal a single-rank render of all samples up to f32 summation order.
"""
import os
import socket
import sys
from pathlib import Path

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = Path(__file__).resolve().parent.parent
W, H, SPP, SEED = 40, 24, 7, 13


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, out_path):
    sys.path.insert(0, str(ROOT))
    sys.path.insert(0, str(ROOT / "tests"))
    import oracle_lib as O
    import raytracing_amd as R
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    first, count = R.shard_samples(SPP, world, rank)
    scene = O.Scene("cornel_box")
    sums = np.zeros((H, W, 3), np.float32)
    if count:
        sums = scene.render(W, H, count, seed=SEED, first_sample=first, threads=2)[1]
    t = torch.from_numpy(np.nan_to_num(sums, nan=0.0))
    dist.reduce(t, dst=0, op=dist.ReduceOp.SUM)
    if rank == 0:
        np.save(out_path, t.numpy())
    dist.barrier()
    dist.destroy_process_group()


def test_two_ranks_reduce_to_the_single_rank_image(tmp_path):
    sys.path.insert(0, str(ROOT / "tests"))
    import oracle_lib as O
    out = tmp_path / "reduced.npy"
    mp.spawn(_worker, args=(2, _free_port(), str(out)), nprocs=2, join=True)
    reduced = np.load(out)
    full = np.nan_to_num(O.Scene("cornel_box").render(W, H, SPP, seed=SEED)[1], nan=0.0)
    assert np.allclose(reduced, full, rtol=1e-5, atol=1e-6)
    assert reduced.sum() > 0
