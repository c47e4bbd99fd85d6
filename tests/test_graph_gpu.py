"""solver.hip_graph: the ascent loop of adversarial_training (adv_compose_solver.py:43-146, 289-405) replayed from a hipGraph.

What must hold:
  * a replay returns what the SAME launch sequence returns when it is enqueued the ordinary way (the frozen launch plan
    without a capture), bit for bit -- loss, parameters, adversarial data;
  * against the ordinary path (kernel selection from this call's own read-backs) the results agree to the tolerance between
    two backward formulations (1e-5 of scale);
  * a replay whose measured displacements leave the intervals of the frozen selection is detected on the device and the
    call is run again the ordinary way: the results are then exactly the ordinary path's;
  * nothing the caller gets back aliases the captured buffers."""
import pytest
import torch

from tests.helpers import make_model, smooth_data

pytestmark = pytest.mark.gpu
DEV = "cuda"

CASES = {
    "2d_full": ((64, 64), ["noise", "bias", "morph", "affine"], 3),
    "2d_morph": ((96, 64), ["morph"], 2),
    "3d_bma": ((32, 32, 16), ["bias", "morph", "affine"], 2),
}


def _solver(dims, names, N, graph):
    import bench
    from advchain_amd.augmentor import AdvAffine, AdvBias, AdvMorph, AdvNoise, ComposeAdversarialTransformSolver
    cls = {"noise": AdvNoise, "bias": AdvBias, "morph": AdvMorph, "affine": AdvAffine}
    chain = [cls[nm](spatial_dims=len(dims), config_dict=cfg, device=torch.device(DEV))
             for nm, cfg in bench.transform_configs(dims, N, names)]
    return ComposeAdversarialTransformSolver(chain_of_transforms=chain, divergence_types=["mse", "contour"],
                                             divergence_weights=[1.0, 0.5], hip_graph=graph)


def _call(solver, data, model, n_iter, seed):
    torch.manual_seed(seed)
    loss = solver.adversarial_training(data=data, model=model, n_iter=n_iter, lazy_load=False, step_sizes=1,
                                       power_iteration=False)
    return ([loss.detach().clone(), solver.adv_data.clone(), solver.warped_back_adv_output.detach().clone(),
             solver.init_output.clone()] + [t.param.detach().clone() for t in solver.chain_of_transforms])


def _deterministic(plan):
    """Does the frozen selection stay on the bit-reproducible formulations?  The window scatter (2D: image warps beyond 16 px,
    squarings beyond 32 px; 3D: beyond 4 voxels) flushes with float atomics: two runs of the SAME launches then differ in the
    last bits, and so do a replay and its launch-by-launch twin."""
    for site in plan.frozen:
        if site["kind"] != "chain":
            continue
        vals, n, d = site["bounds"].values(), site["n"], site["d"]
        lim_sq, lim_warp = (32.0, 16.0) if d == 2 else (4.0, 4.0)
        if any(v >= lim_sq - 0.001 for v in vals[:n]) or vals[n] >= lim_warp - 0.001:
            return False
    return True


def _same(a, b, exact):
    for x, y in zip(a, b):
        if exact:
            assert torch.equal(x, y)
        else:      # (float-atomic window scatter in the plan: summation order, amplified over the steps -- 1.3e-5 seen once in round 6)
            assert float((x - y).abs().max()) <= 3e-5 * max(1e-6, float(y.abs().max())) + 1e-8


def _close(a, b, tol=2e-5):
    for i, (x, y) in enumerate(zip(a, b)):
        scale = max(1e-6, float(y.abs().max()))
        assert float((x - y).abs().max()) <= tol * scale + 1e-7, (i, float((x - y).abs().max()), scale)


@pytest.mark.parametrize("case", sorted(CASES))
@pytest.mark.parametrize("steps", ["one", "many"])
def test_replay_matches_the_ordinary_path(case, steps):
    """One ascent step: the two paths differ by the backward formulation a margin may have selected (<= 1e-4 of scale on what
    one normalised step makes of it).  Several steps: a free-running ascent amplifies that (sign updates, the clamp kinks --
    DESIGN.md section 2), so only the loss is held, loosely; the exact statement is the bit-identity test below."""
    dims, names, n_iter = CASES[case]
    n_iter = 1 if steps == "one" else n_iter
    N = 2
    model = make_model(len(dims), device=DEV)
    eager, graph = _solver(dims, names, N, False), _solver(dims, names, N, True)
    for k in range(6):
        data = smooth_data(N, 1, dims, 40 + k).to(DEV)            # new data every call: the replay copies it in
        want = _call(eager, data, model, n_iter, 100 + k)
        got = _call(graph, data, model, n_iter, 100 + k)
        if steps == "one":
            _close(got, want, 1e-4)
        else:
            _close(got[:1], want[:1], 2e-2)
    st = graph.graph_stats
    assert st["recorded"] == 3 + st["violations"] and st["captures"] >= 1 and st["replays"] >= 2 and st["refused"] == 0, st
    assert eager.graph_stats["replays"] == 0


@pytest.mark.parametrize("case", sorted(CASES))
@pytest.mark.parametrize("steps", ["one", "many"])
def test_replay_is_bit_identical_to_the_same_launches_enqueued_the_ordinary_way(case, steps):
    """The frozen plan without a capture: the same kernels with the same arguments in the same order.  One ascent step keeps
    every displacement on the bit-reproducible formulations (asserted): exact.  Several steps are exact too unless the plan
    reaches the float-atomic window scatter (_deterministic)."""
    from advchain_amd import ops
    dims, names, n_iter = CASES[case]
    n_iter = 1 if steps == "one" else n_iter
    N = 2
    model = make_model(len(dims), device=DEV)
    data = smooth_data(N, 1, dims, 7).to(DEV)
    graph = _solver(dims, names, N, True)
    for k in range(4):
        _call(graph, data, model, n_iter, 300)         # (the same draw as the replays below: inside the plan's intervals)
    (rec,) = graph._graphs.values()
    assert rec["state"] == "replay"
    got = _call(graph, data, model, n_iter, 300)
    assert graph.graph_stats["violations"] == 0
    again = _call(graph, data, model, n_iter, 300)
    exact = _deterministic(rec["plan"])
    assert exact or steps == "many", "one ascent step left the bit-reproducible formulations"
    _same(got, again, exact)
    # the same call on a solver that runs eagerly under the graph's frozen plan
    plain = _solver(dims, names, N, False)
    torch.manual_seed(300)
    plain.init_random_transformation(False)
    flags, steps = [True] * len(names), [1] * len(names)
    plan = rec["plan"]
    plan.rewind()
    plan.flag.zero_()
    ops._PLAN = plan
    try:
        io = plain.get_init_output(data=data, model=model)
        plain.chain_of_transforms = plain.optimizing_transform(data=data, model=model, init_output=io, n_iter=n_iter,
                                                               optimize_flags=flags, step_sizes=steps)
        plan.finish()          # (the premise check of everything the sites measured: one launch, as at the end of a replay)
    finally:
        ops._PLAN = None
    assert plan.cursor == len(plan.frozen) and int(plan.flag.item()) == 0
    _same([t.param.detach() for t in plain.chain_of_transforms], got[4:], exact)
    assert torch.equal(io, got[3])


def test_a_violated_plan_is_detected_and_the_call_runs_the_ordinary_way():
    dims, names, n_iter = CASES["2d_full"]
    n_iter = 2          # (displacements stay below the window scatter's regime: the ordinary path is bit-reproducible)
    N = 2
    model = make_model(2, device=DEV)
    data = smooth_data(N, 1, dims, 9).to(DEV)
    eager, graph = _solver(dims, names, N, False), _solver(dims, names, N, True)
    for k in range(4):
        _call(graph, data, model, n_iter, 400)
    (rec,) = graph._graphs.values()
    assert rec["state"] == "replay" and graph.graph_stats["violations"] == 0
    keep = [site["hi"].clone() for site in rec["plan"].frozen]
    for site in rec["plan"].frozen:          # the graph reads its intervals from these device tensors: shrink them to nothing
        site["hi"].fill_(1e-9)
    want = _call(eager, data, model, n_iter, 500)
    got = _call(graph, data, model, n_iter, 500)
    # detected; this call ran the ordinary way from the same initial parameters -- exactly its results; the graph stays
    assert graph.graph_stats["violations"] == 1 and rec["state"] == "replay" and graph.graph_stats["captures"] == 1
    assert rec["plan"].violated and rec["plan"].violated[0]["bound"] < 1e-8
    for x, y in zip(got, want):
        assert torch.equal(x, y)
    for site, hi in zip(rec["plan"].frozen, keep):
        site["hi"].copy_(hi)
    replays = graph.graph_stats["replays"]
    _call(graph, data, model, n_iter, 400)
    assert graph.graph_stats["violations"] == 1 and graph.graph_stats["replays"] == replays + 1
    # violations in more than one replay of eight: the loop is captured again from the widened record
    for site in rec["plan"].frozen:
        site["hi"].fill_(1e-9)
    for k in range(8):
        _call(graph, data, model, n_iter, 400)
        if rec["state"] != "replay":
            break
    assert rec["state"] == "capture" and rec["recaptures"] == 1
    got = _call(graph, data, model, n_iter, 400)
    assert rec["state"] == "replay" and graph.graph_stats["captures"] == 2
    again = _call(graph, data, model, n_iter, 400)
    _same(got, again, _deterministic(rec["plan"]))


def test_results_do_not_alias_the_captured_buffers():
    dims, names, n_iter = CASES["2d_full"]
    N = 2
    model = make_model(2, device=DEV)
    graph = _solver(dims, names, N, True)
    outs = []
    for k in range(5):
        data = smooth_data(N, 1, dims, 60 + k).to(DEV)
        loss = None
        torch.manual_seed(600 + k)
        loss = graph.adversarial_training(data=data, model=model, n_iter=n_iter, lazy_load=False, step_sizes=1)
        kept = [loss, graph.init_output, graph.adv_data] + [t.param for t in graph.chain_of_transforms]
        outs.append((kept, [x.detach().clone() for x in kept]))
    assert graph.graph_stats["replays"] >= 1
    for kept, copies in outs:                # what an earlier call returned is untouched by the later replays
        for x, y in zip(kept, copies):
            assert torch.equal(x.detach(), y)
    # the returned loss still carries the graph of the final pass (the model's weights train on it)
    model2 = make_model(2, device=DEV).train()
    for p in model2.parameters():
        p.requires_grad_(True)
    g2 = _solver(dims, names, N, True)
    data = smooth_data(N, 1, dims, 70).to(DEV)
    for k in range(5):
        loss = g2.adversarial_training(data=data, model=model2, n_iter=2, lazy_load=False, step_sizes=1)
    assert g2.graph_stats["replays"] >= 1
    loss.backward()
    assert all(p.grad is not None and torch.isfinite(p.grad).all() for p in model2.parameters())


def test_what_a_capture_cannot_hold_takes_the_ordinary_path():
    """Debug prints read device values in the middle of a step: such a solver is refused (and says so in its statistics)."""
    dims, names, n_iter = CASES["3d_bma"]
    N = 2
    model = make_model(3, device=DEV)
    data = smooth_data(N, 1, dims, 11).to(DEV)
    graph = _solver(dims, names, N, True)
    graph.debug = True
    import contextlib
    import io
    for _ in range(3):
        with contextlib.redirect_stdout(io.StringIO()):
            graph.adversarial_training(data=data, model=model, n_iter=1)
    assert graph.graph_stats["refused"] == 3 and graph.graph_stats["replays"] == 0


def _anat_call(solver, data, model, mask, n_iter, seed, tol):
    import contextlib
    import io
    torch.manual_seed(seed)
    out = io.StringIO()
    with contextlib.redirect_stdout(out):
        loss = solver.adversarial_training(data=data, model=model, n_iter=n_iter, lazy_load=False, step_sizes=1,
                                           anatomy_mask_images=mask, anatomy_reg_weight=50, volume_preserve_tolerance=tol)
    return ([loss.detach().clone(), solver.adv_data.clone(), solver.warped_back_adv_output.detach().clone(),
             solver.init_output.clone()] + [t.param.detach().clone() for t in solver.chain_of_transforms]), out.getvalue()


@pytest.mark.parametrize("case", ["3d_morph", "2d_full"])
def test_the_anatomy_ladder_behind_a_replay_passing_check(case):
    """adv_compose_solver.py:369-403 with a mask the field preserves: the n_iter steps (with their regulariser term) and the
    score of the volume check come from the replay, the host only reads the score -- the same messages, the same results
    as the ordinary path to the tolerance between two backward formulations."""
    import bench
    dims, names, n_iter = {"3d_morph": ((32, 32, 16), ["morph"], 2), "2d_full": CASES["2d_full"]}[case]
    n_iter = 1
    N = 2
    model = make_model(len(dims), device=DEV)
    mask = bench.ellipsoid(N, dims).to(DEV)
    eager, graph = _solver(dims, names, N, False), _solver(dims, names, N, True)
    for k in range(6):
        data = smooth_data(N, 1, dims, 60 + k).to(DEV)
        want, said_e = _anat_call(eager, data, model, mask, n_iter, 500 + k, 5e-4)
        got, said_g = _anat_call(graph, data, model, mask, n_iter, 500 + k, 5e-4)
        assert said_e == said_g and "Success" in said_g
        _close(got, want, 1e-4)
    st = graph.graph_stats
    assert st["refused"] == 0 and st["captures"] >= 1 and st["replays"] >= 2 and st["recorded"] == 3 + st["violations"], st


def test_the_anatomy_ladder_behind_a_replay_failing_check():
    """A tolerance nothing meets: the ladder adds steps, draws new parameters at 2 n_iter and gives up at 3 n_iter with a
    fresh draw (adv_compose_solver.py:376-395).  Behind a replay the further steps run the ordinary way: the same number of
    steps, the same messages, the same random draws -- the parameters the call ends with are the ordinary path's, exactly."""
    import bench
    from advchain_amd.augmentor import ComposeAdversarialTransformSolver
    dims, names, n_iter = (32, 32, 16), ["morph"], 2
    N = 2
    model = make_model(3, device=DEV)
    mask = bench.ellipsoid(N, dims).to(DEV)
    eager, graph = _solver(dims, names, N, False), _solver(dims, names, N, True)
    chains = {id(s): list(s.chain_of_transforms) for s in (eager, graph)}
    counts = {id(eager): 0, id(graph): 0}
    inner = ComposeAdversarialTransformSolver._ascent_step

    def counting(self, *a, **k):
        if not torch.cuda.is_current_stream_capturing():
            counts[id(self)] += 1
        return inner(self, *a, **k)
    ComposeAdversarialTransformSolver._ascent_step = counting
    try:
        for k in range(6):
            data = smooth_data(N, 1, dims, 80 + k).to(DEV)
            c0, st0 = dict(counts), dict(graph.graph_stats)
            want, said_e = _anat_call(eager, data, model, mask, n_iter, 700 + k, -1.0)
            got, said_g = _anat_call(graph, data, model, mask, n_iter, 700 + k, -1.0)
            assert said_e == said_g and "new initialization" in said_g and "Success" not in said_g
            assert counts[id(eager)] - c0[id(eager)] == 3 * n_iter
            st1 = graph.graph_stats      # (a replay holds the first n_iter steps; a violated one is run again in full)
            replayed = st1["replays"] > st0["replays"] and st1["violations"] == st0["violations"]
            assert counts[id(graph)] - c0[id(graph)] == (2 * n_iter if replayed else 3 * n_iter), (k, counts, st1)
            # (the reference's ladder leaves the last transform listed twice when it gives up, adv_compose_solver.py:399 -- kept;
            # the caller who goes on with the same solver puts the chain back)
            assert len(got) == len(want) == 4 + len(names) + 1
            for x, y in zip(got[4:], want[4:]):
                assert torch.equal(x, y)
            _close(got[:4], want[:4], 1e-4)
            for s in (eager, graph):
                s.chain_of_transforms = list(chains[id(s)])
    finally:
        ComposeAdversarialTransformSolver._ascent_step = inner
    st = graph.graph_stats
    assert st["refused"] == 0 and st["replays"] >= 2, st


def test_batchnorm_model_in_train_mode_is_captured():
    dims, names, n_iter = CASES["2d_full"]
    N = 2
    torch.manual_seed(0)
    model = torch.nn.Sequential(torch.nn.Conv2d(1, 4, 3, 1, 1), torch.nn.BatchNorm2d(4), torch.nn.ReLU(),
                                torch.nn.Conv2d(4, 4, 1)).to(DEV).train()
    eager, graph = _solver(dims, names, N, False), _solver(dims, names, N, True)
    data = smooth_data(N, 1, dims, 13).to(DEV)
    stats0 = [b.clone() for b in model.buffers()]
    for k in range(6):
        want = _call(eager, data, model, 1, 700 + k)
        stats1 = [b.clone() for b in model.buffers()]
        for b, s in zip(model.buffers(), stats0):
            b.copy_(s)                                            # both solvers see the same running statistics
        got = _call(graph, data, model, 1, 700 + k)
        for b, s in zip(model.buffers(), stats1):                 # and leave the same ones behind
            assert torch.allclose(b.float(), s.float(), rtol=1e-5, atol=1e-7)
        stats0 = [b.clone() for b in model.buffers()]
        _close(got, want, 1e-4)
    assert graph.graph_stats["replays"] >= 2


def test_a_violated_replay_leaves_a_train_mode_model_as_the_ordinary_path_does():
    """the final consistency pass runs the model in train() mode (BatchNorm statistics tracked,
    adv_compose_solver.py:256-259).  With the verdict of a replay only read AFTER that pass, a violated replay ran it twice
    -- once on transforms it then discarded -- and the running statistics / num_batches_tracked saw two updates.  For a
    model with anything in train() mode the verdict is now read first: after a violated call the model's buffers are
    exactly what the ordinary path leaves."""
    dims, names, _ = CASES["2d_full"]
    N, n_iter = 2, 2
    torch.manual_seed(0)
    model = torch.nn.Sequential(torch.nn.Conv2d(1, 4, 3, 1, 1), torch.nn.BatchNorm2d(4), torch.nn.ReLU(),
                                torch.nn.Conv2d(4, 4, 1)).to(DEV).train()
    eager, graph = _solver(dims, names, N, False), _solver(dims, names, N, True)
    data = smooth_data(N, 1, dims, 13).to(DEV)
    for k in range(5):
        _call(graph, data, model, n_iter, 800 + k)
    (rec,) = graph._graphs.values()
    assert rec["state"] == "replay" and graph.graph_stats["violations"] == 0
    stats0 = [b.clone() for b in model.buffers()]
    want = _call(eager, data, model, n_iter, 900)
    stats1 = [b.clone() for b in model.buffers()]
    for b, s in zip(model.buffers(), stats0):
        b.copy_(s)
    for site in rec["plan"].frozen:          # every interval shrunk to nothing: the replay is violated
        site["hi"].fill_(1e-9)
    got = _call(graph, data, model, n_iter, 900)
    assert graph.graph_stats["violations"] == 1
    for b, s in zip(model.buffers(), stats1):
        assert torch.equal(b, s), (b, s)
    for x, y in zip(got, want):
        assert torch.equal(x, y)


def test_hundreds_of_replays_at_the_headline_shape_stay_inside_the_frozen_selection():
    """The fault of LESSONS 66: with the library's clears enqueued as hipMemsetAsync -- memset NODES in the capture -- one
    process in three produced garbage gradients somewhere between the 80th and the 200th replay at this shape (caught by the
    bounds check, so it cost time, not results).  With every node a kernel node, 18 processes x 200 replays were clean; this
    holds one process-worth of that here: no violation, no recapture, finite results."""
    import bench
    wl = bench.WORKLOADS["cfg2"]
    dev = torch.device(DEV)
    torch.manual_seed(1234)
    data = torch.rand(wl["batch"], 1, *wl["dims"], device=dev)
    model = bench.make_model(len(wl["dims"])).to(dev)
    solver = bench.build_solver(wl, dev)
    solver.hip_graph = True
    kw = bench.solver_kwargs(wl, dev)
    for _ in range(solver.hip_graph_record_calls + 200):
        loss = solver.adversarial_training(data=data, model=model, **kw)
    assert torch.isfinite(loss).all() and torch.isfinite(solver.adv_data).all()
    st = solver.graph_stats
    assert st["replays"] >= 195 and st["violations"] == 0 and st["captures"] == 1, st
