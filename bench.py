#!/usr/bin/env python
"""Benchmark of the AdvChain adversarial-augmentation inner loop on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload cfg2]

One "step" = one ``solver.adversarial_training(data, model, n_iter=...)`` call on one synthetic batch (all
ascent steps + the final consistency-loss pass), data already resident in HBM.  Prints ONE JSON line:
metric / value (whole-job augmented images per second) / roofline of the dominant kernel (HIP events on the
launch stream, over the timed region) / cpu_baseline (the CPU oracle, bounded sample, rank 0 at N=1 only).

Multi-GPU (``torch.distributed.run``): one process per GPU, the batch is sharded (weak scaling: the
per-GPU batch is the workload's batch), RCCL all-reduces carry scalars only.
"""
import argparse
import contextlib
import ctypes
import io
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import torch  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6.3 TB/s is the measured achievable

# BASELINE.json configs.  batch = per-GPU batch (weak scaling).
WORKLOADS = {
    "cfg1": dict(dims=(192, 192), batch=4, chain=["noise", "bias", "morph", "affine"], n_iter=1,
                 desc="2D 4x1x192x192, chain=[noise,bias,morph,affine], 1 adv step"),
    "cfg2": dict(dims=(256, 256), batch=32, chain=["noise", "bias", "morph", "affine"], n_iter=5,
                 desc="2D 32x1x256x256, chain=[noise,bias,morph,affine], 5 adv steps"),
    "cfg3": dict(dims=(128, 128, 64), batch=4, chain=["bias", "morph", "affine"], n_iter=3,
                 desc="3D 4x1x128x128x64, chain=[bias,morph,affine], 3 adv steps"),
    "cfg4": dict(dims=(128, 128, 64), batch=8, chain=["noise", "bias", "morph", "affine"], n_iter=5,
                 desc="3D 8x1x128x128x64 per GPU (64 over 8 GPUs), full chain, 5 adv steps"),
    "cfg5": dict(dims=(160, 160, 80), batch=4, chain=["morph"], n_iter=10, anatomy=True,
                 desc="3D 4x1x160x160x80 per GPU (8 over 2 GPUs), morph only (vector h/8), 10 adv steps, anatomy mask"),
    # not BASELINE configs: shapes off the fast paths' alignment (innermost size not a multiple of 4), for profiling
    "odd2d": dict(dims=(250, 250), batch=32, chain=["noise", "bias", "morph", "affine"], n_iter=5,
                  desc="2D 32x1x250x250 (rows not a multiple of 4), full chain, 5 adv steps"),
    "odd3d": dict(dims=(100, 100, 50), batch=4, chain=["bias", "morph", "affine"], n_iter=3,
                  desc="3D 4x1x100x100x50 (rows not a multiple of 4), chain=[bias,morph,affine], 3 adv steps"),
}


def transform_configs(dims, batch, names, morph_div8=False):
    """Notebook conventions."""
    sd = len(dims)
    ds = [batch, 1] + list(dims)
    out = []
    for nm in names:
        if nm == "noise":
            out.append((nm, dict(epsilon=1.0, xi=1e-6, data_size=ds)))
        elif nm == "bias":
            out.append((nm, dict(epsilon=0.3, control_point_spacing=[s // 2 for s in dims],
                                 downscale=2 if sd == 2 else 4, data_size=ds, interpolation_order=3,
                                 init_mode="random", space="log")))
        elif nm == "morph":
            if morph_div8:
                vs = [s // 8 for s in dims]
            elif sd == 2:
                vs = [s // 16 for s in dims]
            else:
                vs = [dims[0] // 16, dims[1] // 16, dims[2] // 2]
            out.append((nm, dict(epsilon=1.5, data_size=ds, vector_size=vs)))
        else:
            if sd == 2:
                out.append((nm, dict(rot=30.0 / 180, scale_x=0.2, scale_y=0.2, shift_x=0.1, shift_y=0.1,
                                     data_size=ds)))
            else:
                out.append((nm, dict(rot_x=10.0 / 180, rot_y=10.0 / 180, rot_z=10.0 / 180, scale_x=0.1,
                                     scale_y=0.1, scale_z=0.1, shift_x=0.1, shift_y=0.1, shift_z=0.1,
                                     data_size=ds)))
    return out


def make_model(sd):
    torch.manual_seed(0)
    conv = torch.nn.Conv2d if sd == 2 else torch.nn.Conv3d
    return conv(1, 4, 3, 1, 1).eval()


def ellipsoid(batch, dims):
    axes = torch.meshgrid([torch.linspace(-1, 1, s) for s in dims], indexing="ij")
    return (sum(a ** 2 for a in axes) <= 0.25 * 1.0).float()[None, None].repeat(batch, 1, *([1] * len(dims)))


# ------------------------------------------------------------------------------------------------
# algorithmic bytes per launch (fp32) of each C-ABI entry point, from its arguments
# ------------------------------------------------------------------------------------------------
def _arr(a, n):
    return [int(a[i]) for i in range(n)]


def _prod(x):
    p = 1
    for v in x:
        p *= v
    return p


def algorithmic_bytes(name, a):
    """Compulsory HBM traffic of one launch: bytes that must be read + written once."""
    if name == "advchain_compose_self_fwd":
        N, nd = a[3], a[4]
        V = _prod(_arr(a[5], nd))
        extra = 4 * nd * N * V if (a[6] & 0xff) == 1 else 0   # final mode also reads phi0 (bits 8..15: displacement hint)
        return 8 * nd * N * V + extra                        # read phi (d ch) + write out (d ch)
    if name == "advchain_compose_self_bwd":
        N, nd = a[6], a[7]
        V = _prod(_arr(a[8], nd))
        return 12 * nd * N * V                               # read grad_out, phi; write grad_phi (atomics)
    if name == "advchain_expo_chain_fwd":                    # n squarings in one call: the sum of their launches
        N, nd, n = a[3], a[4], a[6]
        V = _prod(_arr(a[5], nd))
        return n * 8 * nd * N * V + 4 * nd * N * V
    if name == "advchain_expo_chain_bwd":
        N, nd, n = a[7], a[8], a[10]
        V = _prod(_arr(a[9], nd))
        return n * 12 * nd * N * V
    if name == "advchain_grid_sample_fwd":
        N, C, nd = a[3], a[4], a[5]
        return 4 * N * (C * _prod(_arr(a[6], nd)) + (C + nd) * _prod(_arr(a[7], nd)))
    if name == "advchain_grid_sample_fwd_ride":              # + one rider channel in and out
        N, C, nd = a[5], a[6], a[7]
        return 4 * N * ((C + 1) * _prod(_arr(a[8], nd)) + (C + 1 + nd) * _prod(_arr(a[9], nd)))
    if name == "advchain_affine_warp_fwd_ride":
        N, C, nd = a[5], a[6], a[7]
        return 8 * N * (C + 1) * _prod(_arr(a[8], nd))
    if name == "advchain_grid_sample_bwd":
        N, C, nd = a[6], a[7], a[8]
        IV, OV = _prod(_arr(a[9], nd)), _prod(_arr(a[10], nd))
        b = 4 * N * (C * OV + nd * OV + C * IV)              # grad_out, grid, in
        if a[3]:
            b += 4 * N * C * IV                              # grad_in
        if a[4]:
            b += 4 * N * nd * OV                             # grad_grid
        return b
    if name == "advchain_affine_warp_fwd":
        N, C, nd = a[3], a[4], a[5]
        return 8 * N * C * _prod(_arr(a[6], nd))
    if name == "advchain_affine_warp_bwd":
        N, C, nd = a[6], a[7], a[8]
        V = _prod(_arr(a[9], nd))
        return 4 * N * C * V * (2 + (1 if a[3] else 0))
    if name == "advchain_gauss_axis":
        planes, nd = a[3], a[5]
        return 8 * planes * _prod(_arr(a[6], nd))
    return None


CHAIN_ENTRIES = {"advchain_expo_chain_fwd": ("advchain_compose_self_fwd", 6), "advchain_expo_chain_bwd": ("advchain_compose_self_bwd", 10)}


def launches_in(name, a):
    """Kernel launches of the path's main kernel inside one call of a C-ABI entry (the chain entries run n squarings)."""
    return int(a[CHAIN_ENTRIES[name][1]]) if name in CHAIN_ENTRIES else 1


def entry_label(name, a):
    return name.replace("advchain_", "")


# ------------------------------------------------------------------------------------------------
def build_solver(wl, device, process_group=None, hip_graph=None):
    from advchain_amd.augmentor import (AdvAffine, AdvBias, AdvMorph, AdvNoise,
                                        ComposeAdversarialTransformSolver)
    cls = {"noise": AdvNoise, "bias": AdvBias, "morph": AdvMorph, "affine": AdvAffine}
    sd = len(wl["dims"])
    chain = [cls[nm](spatial_dims=sd, config_dict=cfg, device=device)
             for nm, cfg in transform_configs(wl["dims"], wl["batch"], wl["chain"], morph_div8=wl.get("anatomy", False))]
    solver = ComposeAdversarialTransformSolver(chain_of_transforms=chain, divergence_types=["mse", "contour"],
                                               divergence_weights=[1.0, 0.5], process_group=process_group,
                                               hip_graph=(HIP_GRAPH and (process_group is None or SHARDED_GRAPH) and len(wl["dims"]) == 2)
                                               if hip_graph is None else bool(hip_graph))
    if os.environ.get("ADVCHAIN_PLAN_MARGIN"):       # (experiment knob of tools/sessions/r06_s14.sh)
        solver.hip_graph_margin = float(os.environ["ADVCHAIN_PLAN_MARGIN"])
    if process_group is not None:
        # weak scaling: every rank holds the workload's batch -- the global batch is known without asking the group (saves the
        # all-reduce + host read-back that would otherwise open every call and drain the rank's queue)
        import torch.distributed as dist
        solver.global_batch = wl["batch"] * dist.get_world_size(process_group)
    return solver


def solver_kwargs(wl, device):
    kw = dict(n_iter=wl["n_iter"], step_sizes=1, power_iteration=False)
    if wl.get("anatomy"):
        kw.update(anatomy_mask_images=ellipsoid(wl["batch"], wl["dims"]).to(device), anatomy_reg_weight=50,
                  volume_preserve_tolerance=5e-4)
    return kw


class _ModelTimer(object):
    """HIP events around every call of the user's model inside a step -- the forward passes (get_init_output, every ascent
    step, the final pass) and the input-gradient backward of every ascent step -- on the stream the step runs on.  Used in
    a separate instrumented pass AFTER the timed region (same process, same tensors), so that the timed steps carry no
    extra events: model_ms = GPU time between those events per step; path_ms = step time - model_ms.  The solver calls the
    model through its get_net_output() override point (`model.forward(data)`: module hooks do not fire), so that method is
    wrapped; the backward through the model is bracketed by gradient hooks on its output (fires first) and input (last)."""

    def __init__(self, solver):
        self.solver, self.pairs, self.orig = solver, [], None

    def _event(self):
        e = torch.cuda.Event(enable_timing=True)
        e.record()
        return e

    def __enter__(self):
        self.orig = self.solver.get_net_output

        def timed(model, data):
            e0 = self._event()
            out = self.orig(model, data)
            self.pairs.append((e0, self._event()))
            if torch.is_tensor(out) and out.requires_grad and data.requires_grad:
                box = {}
                out.register_hook(lambda g: box.__setitem__("e0", self._event()))
                data.register_hook(lambda g: self.pairs.append((box["e0"], self._event())) if "e0" in box else None)
            return out
        self.solver.get_net_output = timed
        return self

    def __exit__(self, *exc):
        self.solver.get_net_output = self.orig

    def total_ms(self):
        torch.cuda.synchronize()
        return sum(a.elapsed_time(b) for a, b in self.pairs)


def run_gpu(workload, wl, steps, warmup, rank, world, device):
    """K timed steps of one workload.  Returns (max-over-ranks seconds, roofline of the dominant entry, breakdown,
    extras: model / path split of a step, per-rank step times and collectives per step when sharded)."""
    import torch.distributed as dist
    from advchain_amd import _lib
    pg = dist.group.WORLD if (world > 1 or FORCE_PG) else None
    sd = len(wl["dims"])
    torch.manual_seed(1234 + rank)
    data = torch.rand(wl["batch"], 1, *wl["dims"], device=device)
    model = make_model(sd).to(device)
    solver = build_solver(wl, device, pg)
    kw = solver_kwargs(wl, device)

    import contextlib
    import io

    n_calls = [0]

    def step():
        # the solver prints like the reference does (anatomy ladder messages): keep stdout for the ONE JSON line
        n_calls[0] += 1
        with contextlib.redirect_stdout(io.StringIO()):
            return solver.adversarial_training(data=data, model=model, **kw)

    def sync():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    lib = _lib.load()
    graphed = bool(solver.hip_graph)

    @contextlib.contextmanager
    def ordinary(separate_entries=False):
        """The instrumented passes need the Python-side launches: a replayed graph makes no C-ABI call that could carry events.
        separate_entries: a paired 2D DemonsCompose direction is ONE C call on the ordinary path (advchain_demons_compose_pair_*:
        smoothing, interpolation, chain, bookkeeping) -- events around it would time all of that as one entry; with the
        composite off the same launches are enqueued in the same order by the per-entry calls, which the roofline is about."""
        from advchain_amd import ops as _ops
        composite = _ops.COMPOSITE
        solver.hip_graph = False
        if separate_entries:
            _ops.COMPOSITE = False
        try:
            yield
        finally:
            solver.hip_graph = graphed
            _ops.COMPOSITE = composite
    # warm-up; one warm-up step is instrumented on every entry point to find the dominant kernel -- after one cold step of
    # its own (first launches include code-object loading and would be booked to whichever entry runs first).  With
    # solver.hip_graph the warm-up also holds the calls that record the launch plan and the capture (at least 5 steps).
    dominant = None
    with ordinary():
        step()
        torch.cuda.synchronize()
        lib.records = []
        with lib.timed():
            step()
        torch.cuda.synchronize()
    abi_calls = len(lib.records)           # C-ABI calls of one step on the ordinary path as the product dispatches it
    with ordinary(separate_entries=True):
        lib.records = []
        with lib.timed():
            step()
        torch.cuda.synchronize()
    tot = {}
    for name, a, e0, e1 in lib.records:
        if algorithmic_bytes(name, a) is None:
            continue
        tot[name] = tot.get(name, 0.0) + e0.elapsed_time(e1)
    dominant = max(tot, key=tot.get) if tot else None
    breakdown = {k.replace("advchain_", ""): round(v, 3) for k, v in sorted(tot.items(), key=lambda kv: -kv[1])}
    for i in range(max(warmup - 1, 5 if graphed else 0)):
        step()
    # timed region: exactly K steps between barrier+sync; only the dominant entry point carries events.  The cyclic
    # garbage collector is parked for it, as timeit does: one generation-2 pass (~40-60 ms with torch loaded) would
    # otherwise land inside a 20-step run at random and move the result by 15 %.
    import gc
    gc.collect()
    gc.disable()
    lib.records = []
    n_coll = [0]
    if world > 1 or FORCE_PG:        # collectives per step: every all_reduce the path issues goes through torch.distributed.all_reduce
        orig_all_reduce = dist.all_reduce

        def counted_all_reduce(*a, **k):
            n_coll[0] += 1
            return orig_all_reduce(*a, **k)
        dist.all_reduce = counted_all_reduce
    stats0 = dict(solver.graph_stats)
    from advchain_amd import ops as _ops_mod
    nsteps0 = dict(_ops_mod.NSTEPS_STATS)
    marks = [torch.cuda.Event(enable_timing=True) for _ in range(steps + 1)]      # one event per step boundary (GPU-side span of a step)
    sync()
    t0 = time.perf_counter()
    # ordinary path: one call in five of the dominant entry carries an event pair (every call of a chain entry: it is 8+
    # launches); replayed graphs: nothing can be instrumented inside -- the roofline comes from ordinary steps afterwards
    with lib.timed([dominant] if (dominant and not graphed) else [], every=1 if dominant in CHAIN_ENTRIES else 5):
        marks[0].record()
        for i in range(steps):
            step()
            marks[i + 1].record()
    sync()
    elapsed = time.perf_counter() - t0
    gc.enable()
    extras = {}
    if sd == 3:
        # 3D chains are enqueued with a guessed squaring count (ops._DemonsField: no host read in front of them); a guess
        # that proves wrong enqueues the chain a second time -- how often that happened in the timed region
        extras["step_count_guesses"] = {k: _ops_mod.NSTEPS_STATS[k] - nsteps0[k] for k in ("chains", "respeculated")}
    # GPU-side span of the timed steps: first to last boundary event on the launch stream.  Host-bound steps show up as
    # ms_per_step well above it only at the ends of the region; INSIDE it the two agree by construction, so the honest
    # host-boundness figure is the comparison with the ordinary path below (`ordinary_ms_per_step`) and with rocprof's
    # kernel-busy time (profiles/)
    extras["gpu_span_ms_per_step"] = round(marks[0].elapsed_time(marks[-1]) / steps, 3)
    per = sorted(marks[i].elapsed_time(marks[i + 1]) for i in range(steps))
    extras["gpu_span_ms_by_step"] = {"min": round(per[0], 3), "median": round(per[len(per) // 2], 3), "max": round(per[-1], 3),
                                     "first": [round(marks[i].elapsed_time(marks[i + 1]), 3) for i in range(min(steps, 6))]}
    # kernel-busy time and launches per step: the sum of kernel durations of the rocprofv3 kernel trace of THIS command
    # (tools/profile_bench.sh -> profiles/rNN/per_call_summary.txt, newest round that holds the workload), not an
    # in-process estimate -- a replayed graph carries no per-kernel events.  host_gap = ms_per_step - gpu_busy
    # (round 6: measured in this run -- torch.profiler's kernel records over further calls after the timed region; the
    # committed rocprofv3 summary of the same command only where the profiler is not usable)
    # (one rank only: a profiler that fails on ONE rank of a sharded run would leave the ranks with different numbers of steps,
    # i.e. with collectives that no longer pair up)
    busy = None if (PROFILING_RUN or world > 1) else traced_busy(step, sync)
    if busy is not None and len(busy) == 1:
        extras["gpu_busy_trace"] = busy[0]
        busy = None
    if busy is None:
        busy = profiled_busy(workload)
    if busy is not None:
        extras["gpu_busy_ms_per_step"], extras["launches_per_step"], extras["gpu_busy_source"] = busy
        gap = elapsed / steps * 1e3 - busy[0]
        extras["host_gap_ms_per_step"] = round(max(gap, 0.0), 3)
        if gap < 0:
            # kernel-busy time comes from two FURTHER calls under the profiler (kernels traced one by one run ~1 % longer than
            # in the timed, untraced steps): busy > step means "no measurable idle time", not negative idle time
            extras["host_gap_raw_ms_per_step"] = round(gap, 3)
    extras["abi_calls_per_step"] = abi_calls
    if graphed:
        st = solver.graph_stats
        extras["hip_graph"] = {"replays": st["replays"] - stats0["replays"], "violations": st["violations"] - stats0["violations"],
                               "captures": st["captures"], "recorded_calls": st["recorded"], "refused": st["refused"] - stats0["refused"],
                               "note": "the prediction + the ascent steps of every timed call are ONE hipGraph replay; the final "
                                       "consistency-loss pass is dispatched the ordinary way behind it"}
        viol = [v for rec in solver._graphs.values() for v in rec["plan"].violated]
        if viol:
            extras["hip_graph"]["violated_bounds"] = viol[:8]
        if extras["hip_graph"]["replays"] == 0:
            graphed = False                 # (nothing was replayed -- the anatomy ladder, a capture that failed: the ordinary path was timed)
    k_ord = 0
    if graphed and not PROFILING_RUN:
        # the same steps dispatched launch by launch (what the product does without hip_graph), for comparison
        k_ord = max(3, min(steps, 10))
        with ordinary():
            step()
            sync()
            t1 = time.perf_counter()
            for _ in range(k_ord):
                step()
            sync()
            extras["ordinary_ms_per_step"] = round((time.perf_counter() - t1) / k_ord * 1e3, 3)
    if (not graphed) and (sd == 2 or wl.get("anatomy")) and pg is None and REPLAY_LEG and not PROFILING_RUN:
        # the same workload with the ascent loop of a call replayed from a hipGraph (solver.hip_graph=True: opt-in in the
        # product).  A solver of its own: recorded calls, capture, then K timed replays
        rs = build_solver(wl, device, None, hip_graph=True)

        def rstep():
            with contextlib.redirect_stdout(io.StringIO()):
                return rs.adversarial_training(data=data, model=model, **kw)
        for _ in range(int(rs.hip_graph_record_calls) + 3):
            rstep()
        k_rep = max(3, min(steps, 20))
        r0 = dict(rs.graph_stats)
        sync()
        t1 = time.perf_counter()
        for _ in range(k_rep):
            rstep()
        sync()
        extras["replayed_ms_per_step"] = round((time.perf_counter() - t1) / k_rep * 1e3, 3)
        extras["replayed"] = {"steps": k_rep, "replays": rs.graph_stats["replays"] - r0["replays"],
                              "violations": rs.graph_stats["violations"] - r0["violations"], "captures": rs.graph_stats["captures"],
                              "note": "solver.hip_graph=True (opt-in): prediction + ascent steps of a call as ONE hipGraph replay, the "
                                      "final pass dispatched behind it; `value` / `ms_per_step` are the DEFAULT path (launch by launch)"}
        viol = [v for rec in rs._graphs.values() for v in rec["plan"].violated]
        if viol:
            extras["replayed"]["violated_bounds"] = viol[-8:]
        del rs
    if workload in DETERMINISTIC_LEG and world == 1 and pg is None and not PROFILING_RUN:
        # the same steps with solver.deterministic = True (ordinary path): the window scatter's float-atomic flush replaced by its
        # 64-bit fixed-point twin (bit-reproducible run to run) -- what the mode costs on the workloads that reach that kernel
        from advchain_amd import ops as _ops_det
        ds = build_solver(wl, device, None, hip_graph=False)
        ds.deterministic = True

        def dstep():
            with contextlib.redirect_stdout(io.StringIO()):
                return ds.adversarial_training(data=data, model=model, **kw)
        try:
            for _ in range(2):
                dstep()
            k_det = max(3, min(steps, 10))
            sync()
            t1 = time.perf_counter()
            for _ in range(k_det):
                dstep()
            sync()
            extras["deterministic_ms_per_step"] = round((time.perf_counter() - t1) / k_det * 1e3, 3)
        finally:
            _ops_det.set_deterministic(False)
        del ds
    if workload == "cfg2" and world == 1 and pg is None and not PROFILING_RUN:
        # BASELINE config 2 says "bf16": the USER MODEL under torch.autocast(bfloat16) -- what a mixed-precision training loop
        # hands the solver -- with the augmentation path in fp32 as the reference's own F.grid_sample is under autocast (it is
        # on autocast's fp32 list).  A secondary line: the headline stays the fp32 model (the 1e-4 parity contract is fp32)
        class _Bf16Model(torch.nn.Module):
            def __init__(self, inner):
                super().__init__()
                self.inner = inner

            def forward(self, x):
                with torch.autocast("cuda", dtype=torch.bfloat16):
                    y = self.inner(x)
                return y.float()
        bm = _Bf16Model(model)
        bs = build_solver(wl, device, None, hip_graph=False)

        def bstep():
            with contextlib.redirect_stdout(io.StringIO()):
                return bs.adversarial_training(data=data, model=bm, **kw)
        try:
            for _ in range(3):
                bstep()
            k_bf = max(3, min(steps, 10))
            sync()
            t1 = time.perf_counter()
            for _ in range(k_bf):
                bstep()
            sync()
            extras["model_bf16_autocast_ms_per_step"] = round((time.perf_counter() - t1) / k_bf * 1e3, 3)
        except Exception as exc:        # (a secondary line must not take the headline down with it)
            extras["model_bf16_autocast_error"] = "%s: %s" % (type(exc).__name__, str(exc)[:200])
        del bs, bm
    if not PROFILING_RUN and dominant and not lib.records:
        # nothing in the timed region could carry the dominant entry's events (a replayed graph makes no C-ABI call; the
        # composite DemonsCompose entries hold the chain inside one call): the same launches, enqueued entry by entry
        k_ord = max(3, min(steps, 10))
        with ordinary(separate_entries=True):
            step()
            sync()
            with lib.timed([dominant], every=1 if dominant in CHAIN_ENTRIES else 5):
                for _ in range(k_ord):
                    step()
            sync()
        extras["roofline_steps"] = k_ord
    if world > 1 or FORCE_PG:
        dist.all_reduce = orig_all_reduce
        extras["all_reduces_per_step"] = round(n_coll[0] / float(steps), 2)
    # model / path split: an instrumented pass of a few steps after the timed region (events around the model's forward
    # and backward calls only; the solver's own kernels carry none)
    k_split = 0 if PROFILING_RUN else max(1, min(steps, 5))       # (a profiling run holds the timed workload and nothing else)
    with ordinary(), _ModelTimer(solver) as mt:        # (the timer wraps the Python-side model calls: ordinary steps)
        for _ in range(k_split):
            step()
    model_ms = mt.total_ms() / max(k_split, 1)
    from advchain_amd import ops as _ops
    if _ops.FUSE_STATS["chains"]:
        extras["fused_chain_refusals"] = "%d of %d" % (_ops.FUSE_STATS["refused"], _ops.FUSE_STATS["chains"])
    if k_split:
        extras["model_ms_per_step"] = round(model_ms, 3)
        extras["path_ms_per_step"] = round(elapsed / steps * 1e3 - model_ms, 3)
    extras["model_split_note"] = ("model = HIP-event time of the user model's forward calls and input-gradient backward calls "
                                  "(stock MIOpen / rocBLAS kernels) per adversarial_training call, measured over %d further "
                                  "steps after the timed region; path = ms_per_step - model" % k_split)
    roof = None
    if dominant and lib.records:
        durs, bts, nl = [], [], 0
        for name, a, e0, e1 in lib.records:
            durs.append(e0.elapsed_time(e1) * 1e-3)
            bts.append(algorithmic_bytes(name, a))
            nl += launches_in(name, a)
        # per LAUNCH of the dominant kernel: a chain entry holds n squarings between its two events
        avg_t = sum(durs) / nl
        avg_b = sum(bts) / nl
        achieved = avg_b / avg_t / 1e9
        dominant = CHAIN_ENTRIES.get(dominant, (dominant,))[0]
        traffic, source = profiled_traffic(workload, dominant)
        durs = [0] * nl
        roof = {"timed_in": ("%d steps dispatched entry by entry after the timed region (a replayed graph / a composite C entry "
                             "cannot carry the events of one entry inside it)" % extras.pop("roofline_steps")) if "roofline_steps" in extras
                else "the timed region",
                "bound": "hbm", "kernel": dominant.replace("advchain_", ""), "achieved": round(achieved, 1),
                "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4),
                "traffic": traffic, "traffic_unit": "bytes/launch", "traffic_source": source,
                "launches": len(durs), "avg_launch_us": round(avg_t * 1e6, 2),
                "algorithmic_bytes_per_launch": int(avg_b)}
    extras["solver_calls_in_process"] = n_calls[0]      # (per-call figures of a rocprof summary of this command divide by this)
    if solver.graph_stats["violations"]:              # (each ran its ascent loop a second time, the ordinary way)
        extras["ascent_loops_run_twice"] = solver.graph_stats["violations"]
    if world > 1:
        extras["rank_ms_per_step"] = _rank_spread(elapsed / steps * 1e3, world, device)
        t = torch.tensor([elapsed], device=device, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    return elapsed, roof, breakdown, extras


def traced_busy(step, sync, calls=2):
    """(kernel-busy ms, kernel launches) per call, measured IN THIS RUN: `calls` further steps under torch.profiler's device
    activity (roctracer kernel records: the begin / end timestamps rocprofv3's kernel trace reads), after the timed
    region.  None when the profiler is not usable here."""
    try:
        from torch.profiler import ProfilerActivity, profile
        sync()
        with profile(activities=[ProfilerActivity.CUDA]) as prof:
            for _ in range(calls):
                step()
            sync()
        evs = [e for e in prof.events() if getattr(e, "device_type", None) is not None and "cuda" in str(e.device_type).lower()]
        kern = [e for e in evs if not any(w in e.name.lower() for w in ("memcpy", "memset", "hipmemcpy", "hipmemset"))]
        if not kern:
            return None
        def dur(e):
            return float(getattr(e, "device_time_total", 0.0) or getattr(e, "cuda_time_total", 0.0) or 0.0)
        return (round(sum(dur(e) for e in kern) * 1e-3 / calls, 3), round(len(kern) / float(calls), 1),
                "torch.profiler device activity over %d further calls of this run (kernel records only)" % calls)
    except Exception as exc:        # noqa: BLE001 -- a measurement aid: never takes the bench line down
        return ("unavailable: %s: %s" % (type(exc).__name__, str(exc)[:120]),)


def _rank_spread(ms, world, device):
    """min / max / per-rank ms_per_step over the ranks (load imbalance at a glance; `value` uses the max)."""
    import torch.distributed as dist
    mine = torch.tensor([ms], device=device, dtype=torch.float64)
    every = [torch.zeros_like(mine) for _ in range(world)]
    dist.all_gather(every, mine)
    vals = [round(float(v.item()), 3) for v in every]
    return {"min": min(vals), "max": max(vals), "by_rank": vals}


def profiled_traffic(workload, entry):
    """HBM bytes per launch of `entry` from the committed PMC passes of this command (profiles/rNN/traffic.json, made by
    tools/profile_pmc.sh + tools/traffic_from_pmc.py: rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate runs,
    corrected as profiles/README.md states).  Counters cannot be read from inside the timed run; None if not profiled.
    The newest round that profiled this (workload, entry) wins."""
    root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles")
    try:
        rounds = sorted((d for d in os.listdir(root) if d.startswith("r") and d[1:].isdigit()), reverse=True)
    except OSError:
        return None, None
    for rnd in rounds:
        try:
            with open(os.path.join(root, rnd, "traffic.json")) as f:
                rec = json.load(f)[workload][entry]
            return (rec["traffic_bytes_per_launch"],
                    "profiles/%s/traffic.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes)" % rnd)
        except (OSError, KeyError, ValueError, TypeError):
            continue
    return None, None


def profiled_busy(workload):
    """(kernel-busy ms, kernel launches) per adversarial_training call of `workload` from the newest committed rocprofv3
    kernel-trace summary (profiles/rNN/per_call_summary.txt, line "<workload>_kernel_stats.csv: GPU busy X ms/call, Y
    launches/call"), or None."""
    import re
    root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles")
    try:
        rounds = sorted((d for d in os.listdir(root) if d.startswith("r") and d[1:].isdigit()), reverse=True)
    except OSError:
        return None
    for rnd in rounds:
        try:
            text = open(os.path.join(root, rnd, "per_call_summary.txt")).read()
        except OSError:
            continue
        m = re.search(r"%s_kernel_stats\.csv: GPU busy ([0-9.]+) ms/call, ([0-9.]+) launches/call" % re.escape(workload), text)
        if m:
            return float(m.group(1)), float(m.group(2)), "profiles/%s/per_call_summary.txt (rocprofv3 --kernel-trace --stats of this command)" % rnd
    return None


def _time_pair(x, go, q, halo, reps, disp=None):
    """Average launch duration of grid_sample fwd and bwd on field `q`: `reps` launches back to back between two events
    on the launch stream (one event pair per single launch would add the ~5 us launch gap of an empty queue).  `disp`: the
    measured displacement of q, handed to the forward as the hint the product passes (ops.forward_hint: it picks the forward
    kernel, results do not depend on it); `halo`: the backward's bound from the same measurement (ops.warp_halo)."""
    from advchain_amd import _lib, ops
    lib = _lib.load()
    N = x.shape[0]
    dims = tuple(x.shape[2:])
    gin, ggrid, out = torch.empty_like(x), torch.empty_like(q), torch.empty_like(x)
    ws = ops._scatter_workspace(N, dims, x.device)
    da = _lib.dims_array(dims)
    for _ in range(3):
        ops.raw_grid_sample_fwd(x, q, 0, 0, True, disp_hint=disp)
        ops.raw_grid_sample_bwd(go, x, q, 0, 0, True, True, True, halo)
    cg = 1 | ops._hint_bits(disp)
    ef = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
    torch.cuda.synchronize()
    ef[0].record()
    for _ in range(reps):
        _lib.check(lib.advchain_grid_sample_fwd(ops._ptr(x), ops._ptr(q), ops._ptr(out), N, 1, 3, da, da, 0, 0, cg,
                                                ops._stream()), "fwd")
    ef[1].record()
    for _ in range(reps):
        _lib.check(lib.advchain_grid_sample_bwd(ops._ptr(go), ops._ptr(x), ops._ptr(q), ops._ptr(gin), ops._ptr(ggrid),
                                                ops._ptr(ws), N, 1, 3, da, da, 0, 0, 1, halo, ops._stream()), "bwd")
    ef[2].record()
    torch.cuda.synchronize()
    return ef[0].elapsed_time(ef[1]) * 1e-3 / reps, ef[1].elapsed_time(ef[2]) * 1e-3 / reps


def _trace_pair(x, go, q, halo, reps, disp):
    """Per-kernel durations of the pair from the device's kernel records: `reps` fwd + bwd pairs under torch.profiler.
    Returns {"pair_us", "kernels": {name: avg us}} or None when the profiler is not usable."""
    from advchain_amd import ops
    try:
        from torch.profiler import ProfilerActivity, profile
        torch.cuda.synchronize()
        with profile(activities=[ProfilerActivity.CUDA]) as prof:
            for _ in range(reps):
                ops.raw_grid_sample_fwd(x, q, 0, 0, True, disp_hint=disp)
                ops.raw_grid_sample_bwd(go, x, q, 0, 0, True, True, True, halo)
            torch.cuda.synchronize()
        per = {}
        for e in prof.events():
            if getattr(e, "device_type", None) is None or "cuda" not in str(e.device_type).lower():
                continue
            nm = e.name.split("(")[0]
            if not nm.startswith("void advchain::k_") and not nm.startswith("advchain::k_") and "k_" not in nm:
                continue
            d = float(getattr(e, "device_time_total", 0.0) or getattr(e, "cuda_time_total", 0.0) or 0.0)
            per.setdefault(nm, []).append(d)
        per = {k: v for k, v in per.items() if len(v) >= reps}
        if not per:
            return None
        return {"pair_us": round(sum(sum(v) for v in per.values()) / reps, 2),
                "kernels": {k.replace("void ", "").replace("advchain::", ""): round(sum(v) / len(v), 2) for k, v in per.items()}}
    except Exception:       # noqa: BLE001
        return None


def grid_sample3d_roofline(device, reps=20):
    """North-star kernel: 3D trilinear grid_sample fwd+bwd at 4x1x128x128x64, at TWO displacement levels: a freshly
    initialised AdvMorph field (sub-voxel) and the field the cfg-3 solver ends with after its 3 ascent steps."""
    from advchain_amd import ops
    dims = (128, 128, 64)
    ds = [4, 1] + list(dims)
    nv = 4 * 128 * 128 * 64
    bf, bb = 20 * nv, 36 * nv
    torch.manual_seed(0)
    x = torch.rand(*ds, device=device)
    go = torch.rand(*ds, device=device)
    wl = WORKLOADS["cfg3"]
    solver = build_solver(wl, device)
    morph = [t for t in solver.chain_of_transforms if t.get_name() == "morph"][0]
    levels = {}
    for tag in ("init_field", "after_cfg3_ascent"):
        if tag == "init_field":
            morph.init_parameters()
        else:
            torch.manual_seed(1234)
            data = torch.rand(*ds, device=device)
            solver.adversarial_training(data=data, model=make_model(3).to(device), n_iter=wl["n_iter"], step_sizes=1,
                                        power_iteration=False)
            morph = [t for t in solver.chain_of_transforms if t.get_name() == "morph"][0]
        with torch.no_grad():
            q = morph._field(1.0).contiguous()
        # the displacement bound the product measures in forward (ops._GridSample) and hands to the backward
        entry = ops.grid_displacement(q)
        halo = ops.warp_halo(entry, 3)
        tf, tb = _time_pair(x, go, q, halo, reps, float(entry[1]))
        # the figure to quote: the kernels' own durations from the device's kernel records (what the
        # committed rocprofv3 trace of tools/north_star_pair.py reads, profiles/rNN/ns_pair_summary.txt); the back-to-back
        # event timing overlaps the launch ramp of one kernel with the tail of its twin and reads ~8 % less
        tr = _trace_pair(x, go, q, halo, reps, float(entry[1]))
        t_pair = tr["pair_us"] * 1e-6 if tr else (tf + tb)
        levels[tag] = {"max_displacement_voxels": round(float(entry[1]), 3), "bwd_form": _bwd_form(halo),
                       "frac": round((bf + bb) / t_pair / 1e9 / HBM_PEAK_GBS, 4),
                       "achieved": round((bf + bb) / t_pair / 1e9, 1),
                       "frac_from": "kernel records of this run (torch.profiler device activity)" if tr else "HIP events, launches back to back",
                       "trace": tr,
                       "events": {"fwd_us": round(tf * 1e6, 2), "bwd_us": round(tb * 1e6, 2),
                                  "frac": round((bf + bb) / (tf + tb) / 1e9 / HBM_PEAK_GBS, 4)},
                       "traffic": profiled_traffic("north_star", tag)[0]}
        if tag == "init_field":
            q0 = q
    # second baseline: stock ATen-HIP F.grid_sample forward + backward on the same tensors
    import torch.nn.functional as F
    qn = torch.clamp(q0, -1, 1).permute(0, 2, 3, 4, 1).contiguous()
    xr, qr = x.clone().requires_grad_(True), qn.clone().requires_grad_(True)
    for _ in range(2):
        torch.autograd.grad(F.grid_sample(xr, qr, align_corners=True), (xr, qr), go)
    ea = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    ea[0].record()
    for _ in range(5):
        torch.autograd.grad(F.grid_sample(xr, qr, align_corners=True), (xr, qr), go)
    ea[1].record()
    torch.cuda.synchronize()
    t_aten = ea[0].elapsed_time(ea[1]) * 1e-3 / 5
    worst = min(levels.values(), key=lambda r: r["frac"])
    return {"bound": "hbm", "kernel": "grid_sample3d fwd+bwd @4x1x128x128x64 (C=1, zeros, AdvMorph field)",
            "achieved": worst["achieved"], "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": worst["frac"],
            "frac_is": "the LOWER of the two displacement levels", "levels": levels,
            "algorithmic_bytes": bf + bb, "traffic_unit": "bytes per fwd+bwd pair",
            "aten_hip_fwd_bwd_us": round(t_aten * 1e6, 2), "aten_hip_GBs": round((bf + bb) / t_aten / 1e9, 1)}


def _bwd_form(halo):
    if halo == -1:
        return "gather-form adjoint (z-march), exact bound 1 voxel, one launch"
    if halo < 0:
        return "owner-computes z-march scatter (LDS int32 accumulators), exact bound %d voxels, row-maxima pre-pass + one launch" % -halo
    return "scatter (displacement hint %d voxels)" % halo


def cpu_baseline(wl, name):
    """The CPU oracle (port of the reference's PyTorch-CPU path) on a bounded sample of the same workload, timed with
    ALL host cores as torch threads and, when the host has more than 16, with 16 as well (measured on the GPU box: the
    fastest setting for this path); `value` is the better of the two, `cores` the thread count that produced it."""
    from oracle import advchain_oracle as O
    host = os.cpu_count() or 1
    sd = len(wl["dims"])
    if sd == 2:
        batch, n_iter = min(wl["batch"], 32), wl["n_iter"]   # ~10 s of CPU work at cfg-2
    else:
        batch, n_iter = 1, 1
    cls = {"noise": O.OracleNoise, "bias": O.OracleBias, "morph": O.OracleMorph, "affine": O.OracleAffine}
    # scale to the workload's n_iter: cost ~ (n_iter ascent steps + 1 final pass ~ 0.4 step)
    runs = {}
    for threads in sorted({host, min(host, 16)}):
        # the all-cores leg gets a small sample: on a 256-core host the small ATen ops of this path run ~100x SLOWER
        # with one thread per core than with 16 (measured: 0.036 against 3.8 images/s at cfg-2)
        b, k = (batch, n_iter) if threads <= 16 else (min(batch, 2), 1)
        torch.set_num_threads(threads)
        chain = [cls[nm](sd, cfg) for nm, cfg in transform_configs(wl["dims"], b, wl["chain"],
                                                                    morph_div8=wl.get("anatomy", False))]
        torch.manual_seed(0)
        data = torch.rand(b, 1, *wl["dims"])
        model = make_model(sd)
        solver = O.OracleSolver(chain)
        t0 = time.perf_counter()
        with contextlib.redirect_stdout(io.StringIO()):
            solver.adversarial_training(data=data, model=model, n_iter=k, step_sizes=1)
        dt = time.perf_counter() - t0
        sc = (wl["n_iter"] + 0.4) / (k + 0.4)
        runs[threads] = (round(b / (dt * sc), 4), dt, b, k)
    best = max(runs, key=lambda t: runs[t][0])
    return {"value": runs[best][0], "unit": "images/s" if sd == 2 else "volumes/s", "cores": best, "host_cores": host,
            "threads_used": best, "by_threads": {str(t): v[0] for t, v in runs.items()}, "kind": "port",
            "sample": "%d image(s), %d of %d ascent steps (+ final pass) of %s in %.1f s; scaled linearly in steps"
                      % (runs[best][2], runs[best][3], wl["n_iter"], name, runs[best][1])}


def free_port():
    import socket
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def self_spawn(args, stub):
    """`python bench.py --gpus N` without a launcher: re-exec under torch.distributed.run, one rank per GPU."""
    import subprocess
    if not stub:
        have = torch.cuda.device_count() if torch.cuda.is_available() else 0
        if have < args.gpus:
            sys.stderr.write("bench.py: --gpus %d requested but only %d GPU(s) visible\n" % (args.gpus, have))
            return 3
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


def run_stub(steps, warmup, rank, world):
    """TEST ONLY (ADVCHAIN_BENCH_STUB=1, tests/test_bench_spawn.py): the launcher / rendezvous / reduction skeleton
    of this script on CPU with `gloo` and a step that does nothing, so that the spawn path is covered without a GPU."""
    import torch.distributed as dist
    t0 = time.perf_counter()
    for _ in range(warmup + steps):
        time.sleep(0.001)
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    extras = {}
    if world > 1:
        extras["rank_ms_per_step"] = _rank_spread(elapsed / steps * 1e3, world, torch.device("cpu"))
        extras["all_reduces_per_step"] = 0.0
    t = torch.tensor([elapsed], dtype=torch.float64)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item()), None, {}, extras


PROFILING_RUN = False                   # --only-workload
# Default (round 6): every step dispatched launch by launch from Python -- the product's default path (solver.hip_graph is
# opt-in) is what `value` reports; the 2D workloads are then timed once more with their ascent loop replayed from a hipGraph
# (`replayed_ms_per_step`, `replayed`).  --graph: the replay IS the timed region (round 5's headline), the ordinary path beside
# it (`ordinary_ms_per_step`).  3D workloads never replay here: the host is 1 % of their step, and the replay's safety margin
# moves the 3.2-voxel warps of cfg-3 from the exact 4-voxel march scatter onto the window scatter (14.45 against 14.1 ms)
HIP_GRAPH = False
# sharded runs replay their graphs too (solver._shardable_capture: nothing collective is captured, one all-reduce per call
# checks the premise) only when asked (--sharded-graph): tested with two gloo ranks on one GPU, never on a multi-GPU box
# with RCCL next to a capture, so the default for --gpus N > 1 stays launch by launch
SHARDED_GRAPH = False
REPLAY_LEG = True                       # 2D, one GPU, ordinary path timed: time the replayed ascent loop as well
FORCE_PG = False                        # --force-pg
DETERMINISTIC_LEG = ("cfg2", "cfg5")     # workloads whose backward reaches the window scatter: timed once more in deterministic mode
SECONDARY = ("cfg3", "cfg4", "cfg5")   # the 3D BASELINE configs, timed after the headline workload at N = 1


def result_record(name, wl, world, steps, warmup, elapsed, roof, breakdown, extras=None):
    unit = "images/s" if len(wl["dims"]) == 2 else "volumes/s"
    rec = {"value": round(wl["batch"] * world * steps / elapsed, 3), "unit": unit, "steps": steps, "warmup": warmup,
           "ms_per_step": round(elapsed / steps * 1e3, 3), "workload": "%s: %s" % (name, wl["desc"]),
           "global_batch": wl["batch"] * world, "adv_steps": wl["n_iter"], "roofline": roof,
           "kernel_time_ms_first_step": breakdown}
    ex = dict(extras or {})
    ex.pop("model_split_note", None)
    rec.update(ex)
    return rec


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="cfg2", choices=sorted(WORKLOADS))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--graph", action="store_true",
                    help="time the steps with the ascent loop of a 2D workload replayed from a hipGraph (solver.hip_graph=True, "
                         "opt-in in the product); default: launch by launch -- the product's default path -- with the replay "
                         "timed beside it as `replayed_ms_per_step`")
    ap.add_argument("--no-graph", action="store_true", help="(default since round 6; kept for old command lines)")
    ap.add_argument("--no-replay-leg", action="store_true", help="skip the secondary replayed timing of a 2D workload")
    ap.add_argument("--force-pg", action="store_true",
                    help="--gpus 1: initialise the nccl (= RCCL) process group with ONE rank anyway and run the solver "
                         "through it (device-tensor all-reduces on the data path, `rccl_ranks` from a real all-reduce)")
    ap.add_argument("--sharded-graph", action="store_true",
                    help="--gpus N > 1: replay the 2D ascent loop from a hipGraph on every rank as well (one all-reduce per "
                         "call checks the premise); default for N > 1: launch by launch")
    ap.add_argument("--no-secondary", action="store_true", help="skip the 3D configs timed after the headline workload")
    ap.add_argument("--only-workload", action="store_true",
                    help="profiling runs: the chosen workload and nothing else (no secondary configs, no north-star "
                         "kernel pair, no CPU baseline)")
    args = ap.parse_args()
    global PROFILING_RUN, HIP_GRAPH, SHARDED_GRAPH, REPLAY_LEG
    PROFILING_RUN = bool(args.only_workload)
    HIP_GRAPH = bool(args.graph) and not args.no_graph
    REPLAY_LEG = not (args.no_replay_leg or args.only_workload)
    SHARDED_GRAPH = bool(args.sharded_graph)
    stub = os.environ.get("ADVCHAIN_BENCH_STUB") == "1"
    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        sys.exit(self_spawn(args, stub))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("bench.py: --gpus %d but the launcher started WORLD_SIZE=%d ranks" % (args.gpus, world))
    if stub:
        device = torch.device("cpu")
    else:
        if not torch.cuda.is_available():
            raise SystemExit("bench.py needs a ROCm GPU (no CPU path for the product kernels)")
        if torch.cuda.device_count() <= local_rank:
            raise SystemExit("bench.py: rank %d has no GPU (%d visible)" % (local_rank, torch.cuda.device_count()))
        torch.cuda.set_device(local_rank)
        device = torch.device("cuda", local_rank)
    rccl_ranks = 1
    global FORCE_PG
    FORCE_PG = bool(args.force_pg) and world == 1 and not stub
    if world > 1 or FORCE_PG:
        import torch.distributed as dist
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if stub:
            dist.init_process_group("gloo")
        elif FORCE_PG:       # one rank, no launcher: the rendezvous is this process alone
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", str(free_port()))
            dist.init_process_group("nccl", rank=0, world_size=1, device_id=device)
        else:
            dist.init_process_group("nccl", device_id=device)
        # ranks that actually took part in a collective on device memory (RCCL), not what the environment claims
        one = torch.ones(1, device=device)
        dist.all_reduce(one)
        rccl_ranks = int(round(float(one.item())))
        if rccl_ranks != world:
            raise SystemExit("bench.py: all_reduce saw %d ranks, expected %d" % (rccl_ranks, world))
    wl = WORKLOADS[args.workload]
    if stub:
        elapsed, roof, breakdown, extras = run_stub(args.steps, args.warmup, rank, world)
    else:
        elapsed, roof, breakdown, extras = run_gpu(args.workload, wl, args.steps, args.warmup, rank, world, device)
    if rank == 0:
        rec = result_record(args.workload, wl, world, args.steps, args.warmup, elapsed, roof, breakdown, extras)
        out = {
            "metric": "augmented images/sec (N adv steps, chain=noise+bias+morph+affine)",
            "value": rec["value"], "unit": "images/s", "n_gpus": world, "rccl_ranks": rccl_ranks,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": rec["ms_per_step"], "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": rec["workload"], "global_batch": rec["global_batch"],
                       "adv_steps": wl["n_iter"], "parallelism": "batch-sharded x%d" % world + (" (one-rank RCCL group forced)" if FORCE_PG else ""),
                       "dispatch": "hipGraph replay of the ascent loop (solver.hip_graph)" if (HIP_GRAPH and (world == 1 or SHARDED_GRAPH) and not stub
                                                                                                  and len(wl["dims"]) == 2)
                       else "launch by launch",
                       "segmentation_net": "Conv%dd(1,4,3,1,1) eval (as adv_compose_solver.py:593)" % len(wl["dims"])},
            "roofline": roof,
            "kernel_time_ms_first_step": breakdown,
        }
        out.update(extras)      # model_ms_per_step / path_ms_per_step (+ rank_ms_per_step, all_reduces_per_step when sharded)
        if world == 1 and not stub:
            if not (args.no_secondary or args.only_workload):
                # the 3D configs of BASELINE.json, timed by the same code on the same device (fewer steps: they are
                # 2-15x longer); `value` above stays the headline workload's
                other = {}
                for name in SECONDARY:
                    if name == args.workload:
                        continue
                    w2 = WORKLOADS[name]
                    k2 = max(3, min(args.steps, 5))
                    e2, r2, b2, x2 = run_gpu(name, w2, k2, 1, 0, 1, device)
                    other[name] = result_record(name, w2, 1, k2, 1, e2, r2, b2, x2)
                    other[name].pop("kernel_time_ms_first_step")
                out["other_workloads"] = other
            if not args.only_workload:
                out["roofline_grid_sample3d"] = grid_sample3d_roofline(device)
            if not (args.no_cpu_baseline or args.only_workload):
                out["cpu_baseline"] = cpu_baseline(wl, args.workload)
        print(json.dumps(out))
    if world > 1 or FORCE_PG:
        import torch.distributed as dist
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
