"""Batch sharding over a torch.distributed process group (gloo, world_size 2, CPU): two ranks, each with half of
the batch, must reproduce the single-process whole-batch run -- same updated parameters per sample, same returned
loss -- because every normaliser uses the global batch and only scalars are exchanged."""
import os
import sys
import tempfile

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _specs(sd, n):
    if sd == 2:
        ds = [n, 1, 32, 32]
        return [("noise", dict(epsilon=1.0, xi=1e-6, data_size=ds)),
                ("bias", dict(epsilon=0.3, control_point_spacing=[16, 16], downscale=2, data_size=ds,
                              interpolation_order=3, init_mode="random", space="log")),
                ("morph", dict(epsilon=1.5, data_size=ds, vector_size=[4, 4])),
                ("affine", dict(rot=30 / 180., scale_x=0.2, scale_y=0.2, shift_x=0.1, shift_y=0.1, data_size=ds))]
    ds = [n, 1, 12, 12, 8]
    return [("bias", dict(epsilon=0.3, control_point_spacing=[6, 6, 4], downscale=2, data_size=ds,
                          interpolation_order=3, init_mode="random", space="log")),
            ("morph", dict(epsilon=60.0, data_size=ds, vector_size=[3, 3, 2])),   # big: the 3D step rule kicks in
            ("affine", dict(rot_x=0.05, rot_y=0.05, rot_z=0.05, scale_x=0.1, scale_y=0.1, scale_z=0.1, shift_x=0.1,
                            shift_y=0.1, shift_z=0.1, data_size=ds))]


def _run(sd, data, params, group, if_norm):
    from advchain_amd.augmentor import AdvAffine, AdvBias, AdvMorph, AdvNoise, ComposeAdversarialTransformSolver
    from tests.helpers import make_model
    cls = {"noise": AdvNoise, "bias": AdvBias, "morph": AdvMorph, "affine": AdvAffine}
    chain = [cls[nm](spatial_dims=sd, config_dict=cfg, device=torch.device("cpu")) for nm, cfg in _specs(sd, data.shape[0])]
    for t, p in zip(chain, params):
        t.init_parameters()
        t.set_parameters(p)
    solver = ComposeAdversarialTransformSolver(chain_of_transforms=chain, if_norm_image=if_norm, process_group=group)
    loss = solver.adversarial_training(data=data, model=make_model(sd), n_iter=2, lazy_load=True)
    return float(loss), [t.param.detach().clone() for t in chain], solver.adv_data.detach().clone()


def _inputs(sd):
    from tests.helpers import rand, smooth_data
    from oracle import advchain_oracle as O
    n = 4
    specs = _specs(sd, n)
    data = smooth_data(n, 1, specs[0][1]["data_size"][2:], 77)
    params = []
    for i, (nm, cfg) in enumerate(specs):
        ds = cfg["data_size"]
        if nm == "noise":
            params.append(O.unit_normalize(rand(tuple(ds), 100 + i)))
        elif nm == "bias":
            params.append(0.2 * rand((n, 1) + (4,) * sd, 100 + i))
        elif nm == "morph":
            params.append(O.unit_normalize(rand((n, sd) + tuple(cfg["vector_size"]), 100 + i)))
        else:
            params.append(0.8 * rand((n, 5 if sd == 2 else 9), 100 + i))
    return data, params


def _worker(rank, world, initfile, sd, if_norm, out):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    torch.set_num_threads(2)
    dist.init_process_group("gloo", init_method="file://" + initfile, rank=rank, world_size=world)
    import _pytest.monkeypatch
    from tests import cpu_backend
    mpatch = _pytest.monkeypatch.MonkeyPatch()
    cpu_backend.install(mpatch)
    data, params = _inputs(sd)
    per = data.shape[0] // world
    sl = slice(rank * per, (rank + 1) * per)
    loss, new_params, adv = _run(sd, data[sl].contiguous(), [p[sl].contiguous() for p in params], dist.group.WORLD, if_norm)
    torch.save(dict(loss=loss, params=new_params, adv=adv), os.path.join(out, "rank%d.pt" % rank))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("sd,if_norm", [(2, True), (3, False)])
def test_two_rank_sharding_matches_whole_batch(sd, if_norm, monkeypatch):
    from tests import cpu_backend
    cpu_backend.install(monkeypatch)
    data, params = _inputs(sd)
    ref_loss, ref_params, ref_adv = _run(sd, data, params, None, if_norm)
    with tempfile.TemporaryDirectory() as tmp:
        initfile = os.path.join(tmp, "init")
        mp.spawn(_worker, args=(2, initfile, sd, if_norm, tmp), nprocs=2, join=True)
        parts = [torch.load(os.path.join(tmp, "rank%d.pt" % r)) for r in range(2)]
    # every rank returns the WHOLE-batch loss
    for p in parts:
        assert abs(p["loss"] - ref_loss) < 1e-7 + 1e-5 * abs(ref_loss), (p["loss"], ref_loss)
    for i in range(len(ref_params)):
        got = torch.cat([p["params"][i] for p in parts], dim=0)
        assert float((got - ref_params[i]).abs().max()) < 2e-5, i
    got_adv = torch.cat([p["adv"] for p in parts], dim=0)
    assert float((got_adv - ref_adv).abs().max()) < 5e-5
