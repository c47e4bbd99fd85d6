#!/usr/bin/env python
"""How much the displacements a launch plan records vary from call to call (random initial parameters, fixed data / model):
per chain site (ascent step) the mean, the largest and the smallest of row n-1 (the last squaring's input) over many calls."""
import contextlib, io, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
from advchain_amd import ops

def main(workload="cfg2", calls=200):
    wl = bench.WORKLOADS[workload]; dev = torch.device("cuda")
    torch.manual_seed(1234)
    data = torch.rand(wl["batch"], 1, *wl["dims"], device=dev); model = bench.make_model(len(wl["dims"])).to(dev); kw = bench.solver_kwargs(wl, dev)
    solver = bench.build_solver(wl, dev); solver.hip_graph = True; solver.hip_graph_record_calls = 10 ** 9      # record for ever
    rows = {}
    for c in range(calls):
        with contextlib.redirect_stdout(io.StringIO()):
            solver.adversarial_training(data=data, model=model, **kw)
        (rec,) = solver._graphs.values()
        plan = rec["plan"]
        # the merged record keeps maxima: read this call's own values back from a fresh plan instead
        rec["plan"] = ops.LaunchPlan()
        for i, (kind, r) in enumerate(plan.recorded):
            if kind == "chain":
                rows.setdefault(i, []).append(r["vals"][r["n"] - 1])
    for i, v in sorted(rows.items()):
        t = torch.tensor(v)
        first3 = float(t[:3].max())
        print("site %d: mean %.3f  std %.3f  min %.3f  max %.3f  max/mean %.2f  | max of the first 3 calls %.3f -> calls above 1.3x that: %d of %d, above 1.5x: %d"
              % (i, t.mean(), t.std(), t.min(), t.max(), t.max() / t.mean(), first3, int((t > 1.3 * first3).sum()), len(v), int((t > 1.5 * first3).sum())))

if __name__ == "__main__":
    main(*(sys.argv[1:2]))
