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

from tests.test_dist_gloo import ROOT, _inputs, _specs

pytestmark = pytest.mark.gpu


def _run_gpu(sd, data, params, group, if_norm):
    from advchain_amd.augmentor import AdvAffine, AdvBias, AdvMorph, AdvNoise, ComposeAdversarialTransformSolver
    from tests.helpers import make_model
    dev = torch.device("cuda", 0)
    cls = {"noise": AdvNoise, "bias": AdvBias, "morph": AdvMorph, "affine": AdvAffine}
    chain = [cls[nm](spatial_dims=sd, config_dict=cfg, device=dev) for nm, cfg in _specs(sd, data.shape[0])]
    for t, p in zip(chain, params):
        t.init_parameters()
        t.set_parameters(p.to(dev))
    solver = ComposeAdversarialTransformSolver(chain_of_transforms=chain, if_norm_image=if_norm, process_group=group)
    loss = solver.adversarial_training(data=data.to(dev), model=make_model(sd).to(dev), n_iter=2, lazy_load=True)
    return float(loss), [t.param.detach().cpu() for t in chain], solver.adv_data.detach().cpu()


def _worker(rank, world, initfile, sd, if_norm, out):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    dist.init_process_group("gloo", init_method="file://" + initfile, rank=rank, world_size=world)
    data, params = _inputs(sd)
    per = data.shape[0] // world
    sl = slice(rank * per, (rank + 1) * per)
    loss, new_params, adv = _run_gpu(sd, data[sl].contiguous(), [p[sl].contiguous() for p in params], dist.group.WORLD, if_norm)
    torch.save(dict(loss=loss, params=new_params, adv=adv), os.path.join(out, "rank%d.pt" % rank))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("sd,if_norm", [(2, True), (3, False)])
def test_two_rank_sharding_matches_whole_batch_on_gpu(sd, if_norm):
    data, params = _inputs(sd)
    ref_loss, ref_params, ref_adv = _run_gpu(sd, data, params, None, if_norm)
    with tempfile.TemporaryDirectory() as tmp:
        initfile = os.path.join(tmp, "init")
        mp.spawn(_worker, args=(2, initfile, sd, if_norm, tmp), nprocs=2, join=True)
        parts = [torch.load(os.path.join(tmp, "rank%d.pt" % r)) for r in range(2)]
    for p in parts:   # every rank returns the WHOLE-batch loss
        assert abs(p["loss"] - ref_loss) < 1e-7 + 2e-5 * abs(ref_loss), (p["loss"], ref_loss)
    for i in range(len(ref_params)):
        got = torch.cat([p["params"][i] for p in parts], dim=0)
        assert float((got - ref_params[i]).abs().max()) < 5e-5, i
    got_adv = torch.cat([p["adv"] for p in parts], dim=0)
    assert float((got_adv - ref_adv).abs().max()) < 1e-4
