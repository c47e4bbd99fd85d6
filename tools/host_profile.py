#!/usr/bin/env python
"""Host-side profile (cProfile) of adversarial_training calls of a bench workload: where the Python time goes.

    python tools/host_profile.py [--workload cfg1] [--calls 50]
"""
import argparse
import cProfile
import os
import pstats
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="cfg1")
    ap.add_argument("--calls", type=int, default=50)
    ap.add_argument("--top", type=int, default=45)
    ap.add_argument("--sort", default="tottime")
    ap.add_argument("--inline-backward", action="store_true", help="run the autograd engine on the calling thread, so that the backward functions are profiled too")
    args = ap.parse_args()
    import bench
    if args.inline_backward:
        torch.autograd.set_multithreading_enabled(False)
    wl = dict(bench.WORKLOADS[args.workload])
    dev = torch.device("cuda")
    solver = bench.build_solver(wl, dev)
    torch.manual_seed(0)
    data = torch.rand(wl["batch"], 1, *wl["dims"], device=dev)
    model = bench.make_model(len(wl["dims"])).to(dev)
    kw = bench.solver_kwargs(wl, dev)
    for _ in range(5):
        solver.adversarial_training(data=data, model=model, lazy_load=True, **kw)
    torch.cuda.synchronize()
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(args.calls):
        solver.adversarial_training(data=data, model=model, lazy_load=True, **kw)
    pr.disable()
    torch.cuda.synchronize()
    st = pstats.Stats(pr)
    st.sort_stats(args.sort)
    print("per call: %.3f ms host (profiled)" % (st.total_tt / args.calls * 1e3))
    st.print_stats(args.top)


if __name__ == "__main__":
    main()
