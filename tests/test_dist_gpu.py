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


def _graph_calls(solver, chain, params, data, model, calls, device):
    """`calls` identical calls (the same injected parameters every time): the first ones record the launch plan, then one
    captures, the rest replay.  Returns the result of the last one."""
    import contextlib
    import io
    for _ in range(calls):
        for t, p in zip(chain, params):
            t.init_parameters()
            t.set_parameters(p.to(device))
        solver.chain_of_transforms = list(chain)
        with contextlib.redirect_stdout(io.StringIO()):
            loss = solver.adversarial_training(data=data, model=model, n_iter=2, lazy_load=True)
    return [float(loss), [t.param.detach().cpu().clone() for t in solver.chain_of_transforms[:len(chain)]],
            solver.adv_data.detach().cpu().clone()]


def _graph_solver(data_n, group, device, hip_graph):
    from advchain_amd.augmentor import AdvAffine, AdvBias, AdvMorph, AdvNoise, ComposeAdversarialTransformSolver
    from tests.test_dist_gloo import _specs
    cls = {"noise": AdvNoise, "bias": AdvBias, "morph": AdvMorph, "affine": AdvAffine}
    chain = [cls[nm](spatial_dims=2, config_dict=cfg, device=device) for nm, cfg in _specs(2, data_n)]
    solver = ComposeAdversarialTransformSolver(chain_of_transforms=chain, process_group=group, hip_graph=hip_graph)
    return solver, chain


def _graph_worker(rank, world, initfile, out):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    dist.init_process_group("gloo", init_method="file://" + initfile, rank=rank, world_size=world)
    from tests.helpers import make_model
    device = torch.device("cuda", 0)
    data, params = _inputs(2)
    per = data.shape[0] // world
    sl = slice(rank * per, (rank + 1) * per)
    n_coll = [0]
    orig = dist.all_reduce

    def counted(*a, **k):
        n_coll[0] += 1
        return orig(*a, **k)
    dist.all_reduce = counted
    solver, chain = _graph_solver(per, dist.group.WORLD, device, True)
    model = make_model(2).to(device)
    local = [p[sl].contiguous() for p in params]
    res = _graph_calls(solver, chain, local, data[sl].contiguous().to(device), model, 5, device)
    before = n_coll[0]
    res = _graph_calls(solver, chain, local, data[sl].contiguous().to(device), model, 1, device)
    torch.save(dict(loss=res[0], params=res[1], adv=res[2], stats=dict(solver.graph_stats), collectives=n_coll[0] - before,
                    last_inner=float(solver.last_inner_dist)), os.path.join(out, "rank%d.pt" % rank))
    dist.barrier()
    dist.destroy_process_group()


def test_two_ranks_replaying_their_graphs_match_the_whole_batch():
    """hip_graph under a process group (2D): nothing collective is captured -- the local loss gates the updates inside the
    replay and ONE all-reduce per call (violation flag + per-step losses) checks afterwards that the whole-batch gate would
    have decided the same.  Two ranks on cuda:0, each replaying its own graph, against the single-process whole-batch run
    dispatched the ordinary way."""
    from tests.helpers import make_model
    device = torch.device("cuda", 0)
    data, params = _inputs(2)
    solver, chain = _graph_solver(data.shape[0], None, device, False)
    ref = _graph_calls(solver, chain, params, data.to(device), make_model(2).to(device), 1, device)
    ref_inner = float(solver.last_inner_dist)
    with tempfile.TemporaryDirectory() as tmp:
        initfile = os.path.join(tmp, "init")
        mp.spawn(_graph_worker, args=(2, initfile, tmp), nprocs=2, join=True)
        parts = [torch.load(os.path.join(tmp, "rank%d.pt" % r)) for r in range(2)]
    for p in parts:
        st = p["stats"]
        assert st["captures"] == 1 and st["replays"] >= 2 and st["violations"] == 0 and st["refused"] == 0, st
        # per replayed call: the global batch size, the agreement on the replay state (round 6), the check vector, the final
        # pass's loss -- not one per ascent step
        assert p["collectives"] == 4, p["collectives"]
        assert abs(p["last_inner"] - ref_inner) < 1e-7 + 2e-5 * abs(ref_inner)      # the whole-batch value of the last step
    check_parts(parts, ref, 2e-5, 5e-5, 1e-4)


def _violation_worker(rank, world, initfile, out):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    dist.init_process_group("gloo", init_method="file://" + initfile, rank=rank, world_size=world)
    from tests.helpers import make_model
    device = torch.device("cuda", 0)
    data, params = _inputs(2)
    per = data.shape[0] // world
    sl = slice(rank * per, (rank + 1) * per)
    solver, chain = _graph_solver(per, dist.group.WORLD, device, True)
    model = make_model(2).to(device)
    local, mine = [p[sl].contiguous() for p in params], data[sl].contiguous().to(device)
    _graph_calls(solver, chain, local, mine, model, 5, device)
    (rec,) = solver._graphs.values()
    assert rec["state"] == "replay" and solver.graph_stats["violations"] == 0
    if rank == 0:                                # ONE rank leaves its intervals ...
        for site in rec["plan"].frozen:
            site["hi"].fill_(1e-9)
    res = _graph_calls(solver, chain, local, mine, model, 1, device)
    after_one = dict(solver.graph_stats)
    states = [rec["state"]]
    for _ in range(8):                           # ... again and again: both capture again, in the same call
        _graph_calls(solver, chain, local, mine, model, 1, device)
        states.append(rec["state"])
        if rec["state"] != "replay":
            break
    _graph_calls(solver, chain, local, mine, model, 2, device)
    torch.save(dict(loss=res[0], params=res[1], adv=res[2], after_one=after_one, states=states, final=dict(solver.graph_stats),
                    final_state=rec["state"]), os.path.join(out, "rank%d.pt" % rank))
    dist.barrier()
    dist.destroy_process_group()


def test_a_violation_on_one_rank_sends_every_rank_down_the_ordinary_path():
    """The decision after a replay is taken from all-reduced numbers: a rank whose own replay was fine re-runs the call with
    the rank whose replay was not (their per-step collectives pair up), and both capture again in the same call."""
    from tests.helpers import make_model
    device = torch.device("cuda", 0)
    data, params = _inputs(2)
    solver, chain = _graph_solver(data.shape[0], None, device, False)
    ref = _graph_calls(solver, chain, params, data.to(device), make_model(2).to(device), 1, device)
    with tempfile.TemporaryDirectory() as tmp:
        initfile = os.path.join(tmp, "init")
        mp.spawn(_violation_worker, args=(2, initfile, tmp), nprocs=2, join=True)
        parts = [torch.load(os.path.join(tmp, "rank%d.pt" % r)) for r in range(2)]
    for p in parts:
        assert p["after_one"]["violations"] == 1, p["after_one"]
        assert p["states"] == parts[0]["states"] and p["states"][-1] == "capture", p["states"]
        assert p["final"]["captures"] == 2 and p["final_state"] == "replay", p["final"]
    assert parts[0]["final"] == parts[1]["final"]
    check_parts(parts, ref, 2e-5, 5e-5, 1e-4)


# ------------------------------------------------------------------------------------------------ RCCL itself (round 6)
def _rccl_worker(rank, world, initfile, out):
    """ONE rank on cuda:0 through the `nccl` backend (= RCCL on ROCm): library load, communicator bound to the device
    (`device_id`), all-reduces of DEVICE tensors on the solver's data path -- everything of the multi-GPU path that a one-GPU
    box can execute."""
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    device = torch.device("cuda", 0)
    torch.cuda.set_device(device)
    dist.init_process_group("nccl", init_method="file://" + initfile, rank=rank, world_size=world, device_id=device)
    one = torch.ones(1, device=device)
    dist.all_reduce(one)
    res = {"backend": dist.get_backend(), "ranks": float(one.item())}
    n_coll = [0]
    orig = dist.all_reduce

    def counted(t, *a, **k):
        assert t.is_cuda, "a host tensor reached the RCCL all-reduce"
        n_coll[0] += 1
        return orig(t, *a, **k)
    dist.all_reduce = counted
    for sd, if_norm in ((2, True), (3, False), ("3a", False)):
        data, params = _inputs(sd)
        before = n_coll[0]
        r = _run_gpu(sd, data, params, dist.group.WORLD, if_norm, shard=slice(0, data.shape[0]), global_n=data.shape[0])
        res[str(sd)] = dict(loss=r[0], params=r[1], adv=r[2], scores=r[3] if len(r) > 3 else None, collectives=n_coll[0] - before)
    # the replayed ascent loop under the group (2D): one all-reduce per call carries the check vector
    from tests.helpers import make_model
    data, params = _inputs(2)
    solver, chain = _graph_solver(data.shape[0], dist.group.WORLD, device, True)
    g = _graph_calls(solver, chain, params, data.to(device), make_model(2).to(device), 6, device)
    res["graph"] = dict(loss=g[0], params=g[1], adv=g[2], stats=dict(solver.graph_stats))
    dist.all_reduce = orig
    torch.save(res, os.path.join(out, "rank%d.pt" % rank))
    dist.barrier()
    dist.destroy_process_group()


def test_one_rank_rccl_group_carries_the_solver_collectives():
    """multi-GPU readiness on a one-GPU box.  `dist.init_process_group("nccl", world_size=1,
    device_id=cuda:0)` and one 2D (if_norm_image: min / max all-reduces), one 3D (the step-count norm) and one 3D anatomy
    (ladder decisions) solver call through `process_group`, plus a replayed 2D ascent loop: every collective of the path runs
    through RCCL on device tensors; results equal the run without a group."""
    from tests.helpers import make_model
    refs = {}
    for sd, if_norm in ((2, True), (3, False), ("3a", False)):
        data, params = _inputs(sd)
        refs[str(sd)] = _run_gpu(sd, data, params, None, if_norm)
    device = torch.device("cuda", 0)
    data, params = _inputs(2)
    solver, chain = _graph_solver(data.shape[0], None, device, False)
    gref = _graph_calls(solver, chain, params, data.to(device), make_model(2).to(device), 1, device)
    with tempfile.TemporaryDirectory() as tmp:
        initfile = os.path.join(tmp, "init")
        mp.spawn(_rccl_worker, args=(1, initfile, tmp), nprocs=1, join=True)
        part = torch.load(os.path.join(tmp, "rank0.pt"))
    assert part["backend"] == "nccl" and part["ranks"] == 1.0, part
    for key, ref in refs.items():
        p = part[key]
        assert p["collectives"] >= 2, (key, p["collectives"])
        check_parts([p], ref, 2e-5, 5e-5, 1e-4)
    st = part["graph"]["stats"]
    assert st["captures"] == 1 and st["replays"] >= 2 and st["violations"] == 0, st
    check_parts([part["graph"]], gref, 2e-5, 5e-5, 1e-4)


def test_bench_forced_one_rank_group():
    """`bench.py --gpus 1 --force-pg`: the bench line's `rccl_ranks` comes from a real RCCL all-reduce and the timed steps run
    their collectives through the one-rank group."""
    import json
    import subprocess
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--force-pg", "--workload", "cfg1", "--steps", "3",
                        "--warmup", "1", "--no-secondary", "--no-cpu-baseline", "--no-replay-leg"], capture_output=True, text=True,
                       env=env, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert line["rccl_ranks"] == 1 and line["n_gpus"] == 1 and "forced" in line["config"]["parallelism"], line["config"]
    assert line["all_reduces_per_step"] >= 2, line.get("all_reduces_per_step")
