import io, contextlib, sys
sys.path.insert(0,'.')
import torch
from tests.helpers import Fixture, counted_torch_seed, maxdiff
from tests.test_solver_gpu import make_solver, _kwargs, G6_CASES, DEV
for case in G6_CASES:
    fx = Fixture("g6_" + case)
    solver, chain, meta, model = make_solver(fx)
    with contextlib.redirect_stdout(io.StringIO()), counted_torch_seed(1000):
        loss = solver.adversarial_training(data=fx.t("data", DEV), model=model, **_kwargs(fx, meta))
    ref = fx.f("final_loss")
    errs = dict(loss_rel=abs(float(loss)-ref)/abs(ref), adv=maxdiff(solver.adv_data.cpu(), fx.t("adv_data")),
                wb=maxdiff(solver.warped_back_adv_output.cpu(), fx.t("warped_back")))
    ps=[]
    for i, t in enumerate(solver.chain_of_transforms[:len(chain)]):
        rp = fx.t("final_param_%d" % i)
        ps.append(maxdiff(t.param.cpu(), rp)/max(1.0, float(rp.abs().max())))
    print("%-22s n=%d loss_rel %.1e adv %.1e wb %.1e params %s" % (case, meta["train"]["n_iter"], errs['loss_rel'], errs['adv'], errs['wb'], ["%.1e"%p for p in ps]))
