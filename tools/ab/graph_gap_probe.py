#!/usr/bin/env python
"""Probe: inter-kernel gaps of a hipGraph replay against stream launches (same kernels, same order).

    python tools/ab/graph_gap_probe.py [--n 300]

A chain of n dependent advchain_axpy launches on an 8 MB tensor (2D cfg-2 field size) and a chain of small (4 KB) ones,
timed between two events: eager dispatch from Python, and torch.cuda.graph replay."""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402

from advchain_amd import ops  # noqa: E402


def chain(x, n):
    y = x
    for _ in range(n):
        y = ops.raw_axpy(x, y, 0.5)
    return y


def timed(fn, reps=5):
    best = None
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        e0.record()
        fn()
        e1.record()
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1)
        best = (ms, (t1 - t0) * 1e3) if best is None or ms < best[0] else best
    return best


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=300)
    args = ap.parse_args()
    dev = torch.device("cuda")
    for numel, tag in ((2 * 1024 * 1024, "8 MB"), (1024, "4 KB")):
        x = torch.rand(numel, device=dev)
        chain(x, 8)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            out = chain(x, args.n)
        g.replay()
        torch.cuda.synchronize()
        ref = chain(x, args.n)
        assert torch.equal(ref, out)
        ge, gh = timed(g.replay)
        ee, eh = timed(lambda: chain(x, args.n))
        print("%s x %d launches: eager %.3f ms GPU span (host %.3f ms) = %.2f us/launch | graph %.3f ms (host %.3f ms) = %.2f us/launch"
              % (tag, args.n, ee, eh, ee / args.n * 1e3, ge, gh, ge / args.n * 1e3))




def probe_events_in_graph():
    """Can a pair of timing events recorded INSIDE a capture be read after a replay?"""
    dev = torch.device("cuda")
    x = torch.rand(2 * 1024 * 1024, device=dev)
    chain(x, 4)
    torch.cuda.synchronize()
    for kw in ({"enable_timing": True}, {"enable_timing": True, "external": True}):
        try:
            e0, e1 = torch.cuda.Event(**kw), torch.cuda.Event(**kw)
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                y = chain(x, 10)
                e0.record()
                y = chain(y, 50)
                e1.record()
                y = chain(y, 10)
            for _ in range(3):
                g.replay()
            torch.cuda.synchronize()
            print("events in graph %s: elapsed %.3f ms for 50 launches" % (kw, e0.elapsed_time(e1)))
        except Exception as exc:
            print("events in graph %s: FAILED %s: %s" % (kw, type(exc).__name__, str(exc)[:200]))


if __name__ == "__main__":
    if "--events" in sys.argv:
        sys.argv.remove("--events")
        probe_events_in_graph()
    else:
        main()
