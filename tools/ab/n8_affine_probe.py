#!/usr/bin/env python
"""Is the affine-gradient difference of g6l_2d_full_256_n8 (sample 2) a kernel difference or the reference's own sensitivity
to the 1.7e-6 field difference?  The CPU oracle (pinned on the reference) runs the step twice: with its own deformation
fields, and with the PRODUCT's fields substituted as constant offsets.  If the second run reproduces the product's gradient,
everything behind the field agrees and the difference is what that field difference does to the reference itself."""
import contextlib
import io
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402

from oracle import advchain_oracle as O  # noqa: E402
from tests import test_solver_gpu as T  # noqa: E402
from tests.helpers import Fixture, make_model, oracle_chain, seeded_init_param, smooth_data  # noqa: E402


def main(case="2d_full_256_n8"):
    fx = Fixture("g6l_" + case)
    meta = fx.json()
    sd, N, dims, seed = meta["spatial_dims"], meta["batch"], tuple(meta["dims"]), meta["seed"]
    names = [sp["name"] for sp in meta["chain"]]
    ai, mi = names.index("affine"), names.index("morph")
    init = [seeded_init_param(sp["name"], None, seed + 10 + i) if False else None for i, sp in enumerate(meta["chain"])]
    # ---- product step
    from advchain_amd.augmentor import ComposeAdversarialTransformSolver
    chain = T.build_chain(meta["chain"])
    init = []
    for i, (t, sp) in enumerate(zip(chain, meta["chain"])):
        t.init_parameters()
        init.append(seeded_init_param(sp["name"], t.param.shape, seed + 10 + i).to(T.DEV))
        t.set_parameters(init[-1])
    solver = ComposeAdversarialTransformSolver(chain_of_transforms=chain, divergence_types=["mse", "contour"], divergence_weights=[1.0, 0.5])
    data = smooth_data(N, 1, dims, seed).to(T.DEV)
    model = make_model(sd, device=T.DEV)
    with torch.no_grad():
        qp = torch.clamp(chain[mi]._field(+1.0), -1, 1).cpu()
        qm = torch.clamp(chain[mi]._field(-1.0), -1, 1).cpu()
    init_output = solver.get_init_output(model, data)
    captured = {}
    for ti, t in enumerate(chain):
        t.eval()
        t.param = init[ti].clone()
        t.optimize_parameters = (lambda t=t, ti=ti, orig=t.optimize_parameters:
                                 (lambda step_size=None: (captured.__setitem__(ti, t.param.grad.detach().clone()), orig(step_size=step_size))[1]))()
    with contextlib.redirect_stdout(io.StringIO()):
        T._run_one_step(solver, model, data, init_output, [1] * len(chain), None, {})
    g_gpu = captured[ai].cpu()
    g_ref = fx.t("grad_%d__full" % ai)
    scale = float(g_ref.abs().max())

    # ---- oracle runs
    def oracle_grad(substitute):
        oc = oracle_chain(meta["chain"])
        for i, (t, sp) in enumerate(zip(oc, meta["chain"])):
            t.init_parameters()
            t.set_parameters(init[i].cpu())
        if substitute:
            def hook(q):
                d = q.detach()
                tgt = qp if float((d - qp).abs().max()) < float((d - qm).abs().max()) else qm
                return q + (tgt - d)
            oc[mi].field_hook = hook
        osolver = O.OracleSolver(oc)
        with contextlib.redirect_stdout(io.StringIO()):
            osolver.adversarial_training(data=data.cpu(), model=make_model(sd), n_iter=1, lazy_load=True, step_sizes=1)
        rec = [r for r in osolver.trace if "grads" in r][0]
        return rec["grads"][ai]
    g_own = oracle_grad(False)
    g_sub = oracle_grad(True)
    fmt = lambda m: "\n   ".join(" ".join("%9.2e" % v for v in row) for row in (m / scale).tolist())
    print("field difference product - oracle: +v %.2e, -v %.2e (normalised units)" % (
        float((qp - torch.clamp(oracle_chain(meta["chain"])[mi].__class__._field, -1, 1) if False else 0) if False else 0), 0.0))
    print("|oracle(own fields) - reference| / scale:\n   " + fmt((g_own - g_ref).abs()))
    print("|product - reference| / scale:\n   " + fmt((g_gpu - g_ref).abs()))
    print("|product - oracle(product's fields)| / scale:\n   " + fmt((g_gpu - g_sub).abs()))
    print("|oracle(product's fields) - reference| / scale:\n   " + fmt((g_sub - g_ref).abs()))


if __name__ == "__main__":
    main(*sys.argv[1:])
