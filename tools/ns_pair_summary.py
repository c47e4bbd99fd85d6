#!/usr/bin/env python
"""North-star pair (3D trilinear grid_sample fwd + bwd, 4x1x128x128x64, C = 1) from the rocprofv3 kernel traces of
tools/north_star_pair.py: the advchain kernels of one fwd + bwd pair, their mean durations and the roofline fraction
recomputed from them (234.9 MB algorithmic per pair = 56 B/voxel x 4 194 304 voxels; peak 8 TB/s).

    python tools/ns_pair_summary.py <dir holding ns_{init_field,after_cfg3_ascent}_kernel_stats.csv>
"""
import csv
import os
import sys

ALG_BYTES = 56 * 4 * 128 * 128 * 64
PEAK = 8.0e12
# kernels that belong to the repeated pair (everything else in the trace is set-up: torch.rand, the displacement measurement)
FWD = ("k_sample_march", "k_sample_ring", "k_sample_tiled", "k_grid_sample_fwd", "k_sample_march_flat")
BWD = ("k_adjoint_march", "k_scatter_march3d", "k_march_rowmax", "k_scatter_window3d", "k_scatter_tiled", "k_adjoint_gather",
       "k_overflow", "k_grid_sample_bwd", "k_window")


def main(d):
    for level in ("init_field", "after_cfg3_ascent"):
        path = os.path.join(d, "ns_%s_kernel_stats.csv" % level)
        if not os.path.exists(path):
            print("%s: missing" % path)
            continue
        rows = list(csv.DictReader(open(path)))
        reps = None
        fwd = bwd = 0.0
        lines = []
        for r in rows:
            name = r["Name"].replace("void advchain::", "")
            short = name.split("(")[0]
            kind = "fwd" if short.split("<")[0] in FWD else ("bwd" if any(short.startswith(b) for b in BWD) else None)
            if kind is None:
                continue
            calls = int(r["Calls"])
            if kind == "fwd" and reps is None:
                reps = calls
            lines.append((kind, short, calls, float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e3))
        reps = reps or 1
        for kind, short, calls, avg, tot in lines:
            if kind == "fwd":
                fwd += tot / reps
            else:
                bwd += tot / reps
        pair = fwd + bwd
        print("level %-18s reps %d  fwd %.2f us  bwd %.2f us  pair %.2f us  -> %.3f TB/s = %.3f of %.0f TB/s"
              % (level, reps, fwd, bwd, pair, ALG_BYTES / pair / 1e6, ALG_BYTES / (pair * 1e-6) / PEAK, PEAK / 1e12))
        for kind, short, calls, avg, tot in lines:
            print("    %s %-60s %4d launches  avg %8.2f us" % (kind, short[:60], calls, avg))


if __name__ == "__main__":
    main(sys.argv[1])
