#!/usr/bin/env python
"""The north-star kernel pair (3D trilinear grid_sample fwd + bwd, 4x1x128x128x64, C = 1, zeros padding, clamped AdvMorph
field) on its own, for the PMC passes: the two displacement levels bench.py reports.

    python tools/north_star_pair.py --make-fields /tmp/ns_fields.pt           # not profiled: runs the cfg-3 solver
    rocprofv3 --pmc FETCH_SIZE --kernel-trace ... -- python tools/north_star_pair.py --fields /tmp/ns_fields.pt --level init_field
"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--make-fields")
    ap.add_argument("--fields")
    ap.add_argument("--level", default="init_field", choices=["init_field", "after_cfg3_ascent"])
    ap.add_argument("--reps", type=int, default=5)
    args = ap.parse_args()
    import bench
    from advchain_amd import ops
    dev = torch.device("cuda")
    ds = [4, 1, 128, 128, 64]
    if args.make_fields:
        wl = bench.WORKLOADS["cfg3"]
        solver = bench.build_solver(wl, dev)
        morph = [t for t in solver.chain_of_transforms if t.get_name() == "morph"][0]
        torch.manual_seed(0)
        morph.init_parameters()
        with torch.no_grad():
            q0 = morph._field(1.0).contiguous().cpu()
        torch.manual_seed(1234)
        data = torch.rand(*ds, device=dev)
        solver.adversarial_training(data=data, model=bench.make_model(3).to(dev), n_iter=wl["n_iter"], step_sizes=1,
                                    power_iteration=False)
        morph = [t for t in solver.chain_of_transforms if t.get_name() == "morph"][0]
        with torch.no_grad():
            q1 = morph._field(1.0).contiguous().cpu()
        torch.save({"init_field": q0, "after_cfg3_ascent": q1}, args.make_fields)
        return
    q = torch.load(args.fields)[args.level].to(dev)
    torch.manual_seed(0)
    x, go = torch.rand(*ds, device=dev), torch.rand(*ds, device=dev)
    entry = ops.grid_displacement(q)
    halo = ops.warp_halo(entry, 3)
    for _ in range(args.reps):
        ops.raw_grid_sample_fwd(x, q, 0, 0, True, disp_hint=entry[1])
        ops.raw_grid_sample_bwd(go, x, q, 0, 0, True, True, True, halo)
    torch.cuda.synchronize()
    print("level %s displacement %.3f halo %d reps %d" % (args.level, entry[1], halo, args.reps))


if __name__ == "__main__":
    main()
