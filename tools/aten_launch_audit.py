#!/usr/bin/env python
"""Which host lines launch non-advchain (ATen / rocclr) kernels during one adversarial_training call?

    python tools/aten_launch_audit.py [--workload cfg2]

Runs the bench workload under torch.profiler with stacks and prints, per ATen op that launched a GPU kernel, the count per
call and the innermost frame inside advchain_amd/ (or bench.py) that issued it.  The user's model (conv) is reported apart.
"""
import argparse
import collections
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from torch.profiler import ProfilerActivity, profile  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="cfg2")
    ap.add_argument("--calls", type=int, default=2)
    args = ap.parse_args()
    import bench
    wl = dict(bench.WORKLOADS[args.workload])
    dev = torch.device("cuda")
    solver = bench.build_solver(wl, dev)
    torch.manual_seed(0)
    data = torch.rand(wl["batch"], 1, *wl["dims"], device=dev)
    model = bench.make_model(len(wl["dims"])).to(dev)
    kw = bench.solver_kwargs(wl, dev)
    for _ in range(2):
        solver.adversarial_training(data=data, model=model, lazy_load=True, **kw)
    torch.cuda.synchronize()
    # 1. main-thread attribution: every ATen op dispatched from Python, with the innermost advchain_amd / bench frame
    import traceback
    from torch.utils._python_dispatch import TorchDispatchMode
    sites = collections.Counter()

    class Audit(TorchDispatchMode):
        def __torch_dispatch__(self, func, types, a=(), kw=None):
            name = str(func)
            if not any(k in name for k in ("empty", "view", "detach", "reshape", "_to_copy.default_x", "alias", "expand", "slice",
                                           "select", "permute", "as_strided", "t.default", "unsqueeze", "squeeze", "_unsafe_view",
                                           "convolution", "is_", "stride", "size", "numel", "dim", "record_stream", "lift_fresh")):
                fr = [f for f in traceback.extract_stack() if "advchain_amd" in f.filename or f.filename.endswith("bench.py")]
                site = "%s:%d %s" % (os.path.basename(fr[-1].filename), fr[-1].lineno, fr[-1].line) if fr else "?"
                sites[(name, site)] += 1
            return func(*a, **(kw or {}))
    with Audit():
        for _ in range(args.calls):
            solver.adversarial_training(data=data, model=model, lazy_load=True, **kw)
        torch.cuda.synchronize()
    print("== ATen ops dispatched from the main thread, per call (%s) [backward host code runs on the autograd thread and is"
          " not seen here]" % args.workload)
    tot = 0
    for (name, site), n in sorted(sites.items(), key=lambda kv: -kv[1]):
        print("%6.1f  %-28s %s" % (n / args.calls, name[:28], site[:120]))
        tot += n
    print("total %.1f per call" % (tot / args.calls))
    # 2. kernel counts per ATen op (all threads)
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
        for _ in range(args.calls):
            solver.adversarial_training(data=data, model=model, lazy_load=True, **kw)
        torch.cuda.synchronize()
    by_op = collections.Counter()
    for ev in prof.events():
        if ev.device_type != torch.autograd.DeviceType.CPU or not ev.kernels:
            continue
        by_op[ev.name] += len(ev.kernels)
    print("== GPU kernels per op and call (custom Functions include the advchain kernels they launch)")
    print("   " + ", ".join("%s %.1f" % (k, v / args.calls) for k, v in sorted(by_op.items(), key=lambda kv: -kv[1])))


if __name__ == "__main__":
    main()
