#!/usr/bin/env python
"""Violation rate and step time of the replayed ascent loop of a bench workload over many processes-worth of captures
(the stress that exposed the memset-node fault of LESSONS 66): graph_margin_ab.py [workload] [runs]."""
import contextlib, io, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
from advchain_amd import ops

def run(tag, margin, records, steps=300, workload="cfg2"):
    wl = bench.WORKLOADS[workload]; dev = torch.device("cuda")
    torch.manual_seed(int(os.environ.get("AB_SEED", "1234")))
    data = torch.rand(wl["batch"], 1, *wl["dims"], device=dev); model = bench.make_model(len(wl["dims"])).to(dev); kw = bench.solver_kwargs(wl, dev)
    init = ops.LaunchPlan.__init__
    ops.LaunchPlan.__init__ = lambda self, m=margin: init(self, m)
    try:
        solver = bench.build_solver(wl, dev); solver.hip_graph = True; solver.hip_graph_record_calls = records

        def step():
            with contextlib.redirect_stdout(io.StringIO()):
                return solver.adversarial_training(data=data, model=model, **kw)
        for _ in range(records + 2): step()
        torch.cuda.synchronize(); v0 = solver.graph_stats["violations"]; t0 = time.perf_counter()
        for _ in range(steps): step()
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / steps * 1e3
        print("%-28s %.3f ms/call, %d violations in %d replays, captures %d" % (tag, dt, solver.graph_stats["violations"] - v0, steps, solver.graph_stats["captures"]), flush=True)
        for rec in solver._graphs.values():
            for v in rec["plan"].violated[:12]:
                print("      ", v)
    finally:
        ops.LaunchPlan.__init__ = init

if __name__ == "__main__":
    w = sys.argv[1] if len(sys.argv) > 1 else "cfg2"
    runs = int(sys.argv[2]) if len(sys.argv) > 2 else 4
    if os.environ.get("AB_NO_COMPOSITE"):
        ops.COMPOSITE = False
    if os.environ.get("AB_NO_FUSE"):
        ops.FUSE_2D = False
    margins = [float(m) for m in os.environ.get("AB_MARGINS", "1.3").split(",")]
    for i in range(runs):
        for m in margins:
            run("flat %.2f, 3 records" % m, m, 3, steps=int(os.environ.get("AB_STEPS", "200")), workload=w)
