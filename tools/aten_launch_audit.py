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
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
        for _ in range(args.calls):
            solver.adversarial_training(data=data, model=model, lazy_load=True, **kw)
        torch.cuda.synchronize()
    by_site = collections.Counter()
    model_ops = collections.Counter()
    for ev in prof.events():
        if ev.device_type != torch.autograd.DeviceType.CPU or not ev.kernels:
            continue
        frames = [f for f in (ev.stack or []) if "advchain_amd" in f or "bench.py" in f]
        site = frames[0].strip() if frames else "(no advchain frame)"
        n = len(ev.kernels)
        if "conv" in ev.name or "miopen" in ev.name.lower():
            model_ops[ev.name] += n
        else:
            by_site[(ev.name, site)] += n
    tot = 0
    print("== ATen ops with GPU kernels, per call (%s)" % args.workload)
    for (name, site), n in sorted(by_site.items(), key=lambda kv: -kv[1]):
        print("%6.1f  %-34s %s" % (n / args.calls, name[:34], site[-110:]))
        tot += n
    print("total %.1f per call; model conv ops: %s" % (tot / args.calls, {k: v / args.calls for k, v in model_ops.items()}))


if __name__ == "__main__":
    main()
