"""Batch sharding with the REAL kernels: two ranks on cuda:0 (gloo carries the scalar all-reduces; NCCL refuses two
ranks on one device), each with half of the batch, against the single-process whole-batch run on the same GPU.
Same protocol as tests/test_dist_gloo.py, which checks the host logic on CPU."""
import os
import sys
import tempfile

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from tests.test_dist_gloo import ROOT, _inputs, _run, check_parts

pytestmark = pytest.mark.gpu


def _run_gpu(sd, data, params, group, if_norm, shard=None, global_n=None):
    return _run(sd, data, params, group, if_norm, device=torch.device("cuda", 0), shard=shard, global_n=global_n)


def _worker(rank, world, initfile, sd, if_norm, out):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    dist.init_process_group("gloo", init_method="file://" + initfile, rank=rank, world_size=world)
    data, params = _inputs(sd)
    per = data.shape[0] // world
    sl = slice(rank * per, (rank + 1) * per)
    res = _run_gpu(sd, data[sl].contiguous(), [p[sl].contiguous() for p in params], dist.group.WORLD, if_norm, shard=sl,
                   global_n=data.shape[0])
    torch.save(dict(loss=res[0], params=res[1], adv=res[2], scores=res[3] if len(res) > 3 else None),
               os.path.join(out, "rank%d.pt" % rank))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("sd,if_norm", [(2, True), (3, False), ("3a", False)])
def test_two_rank_sharding_matches_whole_batch_on_gpu(sd, if_norm):
    data, params = _inputs(sd)
    ref = _run_gpu(sd, data, params, None, if_norm)
    with tempfile.TemporaryDirectory() as tmp:
        initfile = os.path.join(tmp, "init")
        mp.spawn(_worker, args=(2, initfile, sd, if_norm, tmp), nprocs=2, join=True)
        parts = [torch.load(os.path.join(tmp, "rank%d.pt" % r)) for r in range(2)]
    check_parts(parts, ref, 2e-5, 5e-5, 1e-4)
