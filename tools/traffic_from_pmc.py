#!/usr/bin/env python
"""HBM-side traffic per C-ABI entry launch from the rocprofv3 PMC summaries (tools/profile_pmc.sh -> pmc_summary.py).

FETCH_SIZE / WRITE_SIZE are reported in KB.  Calibration on this rocprofv3/gfx950 (profiles/README.md, following
MI355X_MICROARCH.md "HBM"): FETCH_SIZE counts 1/2 of the bytes of 16-B-per-lane streaming reads (x2.0) and ~0.84 of
4-B-per-lane coalesced reads (x1.19); WRITE_SIZE is 1:1.  The factor per kernel is chosen by how that kernel loads.

usage: traffic_from_pmc.py <dir with pmc_{fetch,write}_size_<workload>.csv> <out.json>
"""
import csv
import json
import os
import re
import sys

ENTRIES = {   # entry -> (regexes of the kernels an entry launch runs, regex of its MAIN kernel (one per launch))
    "advchain_grid_sample_bwd": ([r"k_scatter_rows<\d, \d, \d, false", r"k_scatter_overflow<\d, 0>", r"k_gather_overflow<\d, 0>",
                                  r"k_adjoint_gather<\d, \d, \d, false", r"k_grid_sample_bwd<", r"k_scatter_window[23]d<\d, \d, false"],
                                 r"k_scatter_rows<\d, \d, \d, false|k_adjoint_gather<\d, \d, \d, false|k_grid_sample_bwd<|k_scatter_window[23]d<\d, \d, false"),
    "advchain_compose_self_bwd": ([r"k_scatter_rows<\d, 1, \d, true", r"k_scatter_overflow<\d, 1>", r"k_gather_overflow<\d, 1>",
                                   r"k_adjoint_gather<\d, \d, \d, true", r"k_compose_self_bwd<", r"k_scatter_window[23]d<\d, \d, true"],
                                  r"k_scatter_rows<\d, 1, \d, true|k_adjoint_gather<\d, \d, \d, true|k_compose_self_bwd<|k_scatter_window[23]d<\d, \d, true"),
    "advchain_compose_self_fwd": ([r"k_compose_self_fwd<", r"k_sample_tiled<\d, 1, \d, true>", r"k_sample_march<3, true"],
                                  r"k_compose_self_fwd<|k_sample_tiled<\d, 1, \d, true>|k_sample_march<3, true"),
    "advchain_grid_sample_fwd": ([r"k_grid_sample_fwd<", r"k_sample_tiled<\d, \d, \d, false>", r"k_sample_march<\d, false"],
                                 r"k_grid_sample_fwd<|k_sample_tiled<\d, \d, \d, false>|k_sample_march<\d, false"),
    "advchain_affine_warp_fwd": ([r"k_affine_warp_fwd<"], r"k_affine_warp_fwd<"),
    "advchain_affine_warp_bwd": ([r"k_affine_warp_bwd<", r"k_affine_gather_bwd<", r"k_affine_geometry<", r"k_reduce_partials"],
                                 r"k_affine_warp_bwd<"),
    "advchain_gauss_axis": ([r"k_gauss_axis", r"k_gauss_march_z", r"k_gauss_xy"], r"k_gauss_axis|k_gauss_march_z|k_gauss_xy"),   # per launch of any pass
    "advchain_tp_interp_fwd": ([r"k_tp_interp_fwd<"], r"k_tp_interp_fwd<"),
    "advchain_band_reduce_axis": ([r"k_band_reduce"], r"k_band_reduce"),
    "advchain_bias_field_fwd": ([r"k_bias_field_fwd|k_bias_fwd"], r"k_bias_field_fwd|k_bias_fwd"),
    "advchain_bias_field_bwd": ([r"k_bias_field_bwd|k_bias_bwd"], r"k_bias_field_bwd|k_bias_bwd"),
    "advchain_consistency_fwd": ([r"k_softmax_diff", r"k_edge_fwd"], r"k_softmax_diff"),
    "advchain_consistency_bwd": ([r"k_consistency_bwd"], r"k_consistency_bwd"),
    "advchain_axpy": ([r"k_axpy"], r"k_axpy"),
}
# kernels whose global reads are 16 B per lane (FETCH_SIZE x 2.0); the rest read 4 B per lane (x 1.19).  The marching
# forward sampler mixes both (16-B staging of the image, 4-B grid loads): its factor is the blend for its byte split.
for _e in ("advchain_grid_sample_bwd", "advchain_compose_self_bwd"):
    ENTRIES[_e][0].append(r"k_adjoint_march<")
    ENTRIES[_e] = (ENTRIES[_e][0], ENTRIES[_e][1] + r"|k_adjoint_march<")
ENTRIES["advchain_grid_sample_bwd"] = ([p for p in ENTRIES["advchain_grid_sample_bwd"][0] if p != r"k_adjoint_march<"] +
                                       [r"k_adjoint_march<\d, false"],
                                       ENTRIES["advchain_grid_sample_bwd"][1].replace(r"|k_adjoint_march<", r"|k_adjoint_march<\d, false"))
ENTRIES["advchain_compose_self_bwd"] = ([p for p in ENTRIES["advchain_compose_self_bwd"][0] if p != r"k_adjoint_march<"] +
                                        [r"k_adjoint_march<3, true"],
                                        ENTRIES["advchain_compose_self_bwd"][1].replace(r"|k_adjoint_march<", r"|k_adjoint_march<3, true"))
# round-2 owner-computes scatters (scatter_march.hip): 3D z-march, 2D whole rows, and their row-maxima pre-pass
for _e, _self in (("advchain_grid_sample_bwd", "false"), ("advchain_compose_self_bwd", "true")):
    _p = [r"k_scatter_march3d<\d, \d, %s" % _self, r"k_scatter_rows2d<\d, \d, %s" % _self]
    ENTRIES[_e] = (ENTRIES[_e][0] + _p, ENTRIES[_e][1] + "|" + "|".join(_p))
ENTRIES["advchain_grid_sample_bwd"][0].append(r"k_march_rowmax(64)?<[14]>")
ENTRIES["advchain_compose_self_bwd"][0].append(r"k_march_rowmax(64)?<[23]>")
# round 3: the 16-byte form of the 3D march scatter for image warps, the ring forward for 2..4-voxel fields
ENTRIES["advchain_grid_sample_bwd"] = (ENTRIES["advchain_grid_sample_bwd"][0] + [r"k_scatter_march3d_wide<"],
                                       ENTRIES["advchain_grid_sample_bwd"][1] + r"|k_scatter_march3d_wide<")
ENTRIES["advchain_grid_sample_fwd"] = (ENTRIES["advchain_grid_sample_fwd"][0] + [r"k_sample_ring<"],
                                       ENTRIES["advchain_grid_sample_fwd"][1] + r"|k_sample_ring<")
ENTRIES["advchain_affine_warp_fwd"] = ([r"k_affine_warp_fwd", r"k_affine_box_fwd"], r"k_affine_warp_fwd|k_affine_box_fwd")
# round 3: LDS-box theta gradient and owner-computes grad_in scatter; one entry launch = one k_affine_geometry-less count of
# the theta kernel (every backward call launches exactly one k_reduce_partials)
ENTRIES["advchain_affine_warp_bwd"] = ([r"k_affine_warp_bwd<", r"k_affine_gather_bwd<", r"k_affine_geometry<", r"k_reduce_partials",
                                        r"k_affine_box_gtheta<", r"k_affine_box_gin<"], r"k_reduce_partials")
# round 3, second half: rows of 68 .. 128 voxels -- flat forward march, flat march scatter of the self-composition (the wide
# adjoint march keeps the name k_adjoint_march, with a seventh template argument)
ENTRIES["advchain_compose_self_fwd"] = (ENTRIES["advchain_compose_self_fwd"][0] + [r"k_sample_march_flat<3, true"],
                                        ENTRIES["advchain_compose_self_fwd"][1] + r"|k_sample_march_flat<3, true")
ENTRIES["advchain_grid_sample_fwd"] = (ENTRIES["advchain_grid_sample_fwd"][0] + [r"k_sample_march_flat<\d, false"],
                                       ENTRIES["advchain_grid_sample_fwd"][1] + r"|k_sample_march_flat<\d, false")
ENTRIES["advchain_compose_self_bwd"] = (ENTRIES["advchain_compose_self_bwd"][0] + [r"k_scatter_march3d_flat<"],
                                        ENTRIES["advchain_compose_self_bwd"][1] + r"|k_scatter_march3d_flat<")
# round 4: the sub-pixel squarings of a 2D chain in one launch each way, the dense-band upsample adjoint, the fused 2D loss
ENTRIES["advchain_compose_self_fwd"] = (ENTRIES["advchain_compose_self_fwd"][0] + [r"k_expo_fused_fwd2d<", r"k_compose_self_fwd_gated<"],
                                        ENTRIES["advchain_compose_self_fwd"][1])
ENTRIES["advchain_compose_self_bwd"] = (ENTRIES["advchain_compose_self_bwd"][0] + [r"k_adjoint_fused2d<"], ENTRIES["advchain_compose_self_bwd"][1])
ENTRIES["advchain_consistency_fwd"] = ([r"k_softmax_diff", r"k_edge_fwd", r"k_loss_fused_fwd4"], r"k_softmax_diff|k_loss_fused_fwd4")
ENTRIES["advchain_consistency_bwd"] = ([r"k_consistency_bwd", r"k_loss_fused_bwd4"], r"k_consistency_bwd|k_loss_fused_bwd4")
# A fused launch covers several squarings, so the launches of the chain entries are counted per SQUARING, as bench.py does
# (its roofline divides a chain call by n): chain calls x 8, the chain calls read off the one Gaussian launch that opens the
# backward of a chain (k_gauss_xy<0, .>) / closes its forward (k_gauss_xy<2, .>)
CHAIN_CALLS = {"advchain_compose_self_fwd": r"k_gauss_xy<2, \d>", "advchain_compose_self_bwd": r"k_gauss_xy<0, \d>"}
CHAIN_N = 8
# algorithmic bytes of one squaring on the PAIRED batch [v; -v] (forward 8 d, backward 12 d bytes per voxel)
PAIRED_VOXELS = {"cfg2": (2, 64 * 256 * 256), "cfg3": (3, 8 * 128 * 128 * 64), "cfg4": (3, 16 * 128 * 128 * 64), "cfg5": (3, 8 * 160 * 160 * 80)}
HBM_PEAK = 8.0e12
WIDE = re.compile(r"k_expo_fused_fwd2d|k_adjoint_fused2d|k_band_reduce_rows|k_loss_fused|k_sample_march_flat<3, true|"r"k_adjoint_gather|k_adjoint_march|k_sample_tiled|k_sample_march<3, true|k_gauss_axis_v4|k_gauss_march_z|k_gauss_xy|k_max_displacement|"
                  r"k_axpy|k_absmax|k_softmax_diff_v4|k_edge_fwd_march4|k_consistency_bwd_march4|k_affine_box_fwd|k_affine_box_gtheta|"
                  r"k_scatter_march3d_wide|k_march_rowmax64|k_sample_ring")   # 16 B / lane
MIXED = {r"k_sample_march<1, false": 1.41, r"k_sample_march<4, false": 1.7,
         r"k_sample_march_flat<1, false": 1.41, r"k_sample_march_flat<4, false": 1.7}   # image 16 B/lane + 3 grid channels 4 B/lane


def read_factor(kernel):
    for pat, f in MIXED.items():
        if re.search(pat, kernel):
            return f
    return 2.0 if WIDE.search(kernel) else 1.0 / 0.84


def load(path):
    rows = {}
    if os.path.exists(path):
        for r in csv.DictReader(open(path)):
            rows[r["kernel"]] = (int(r["calls"]), float(r["mean_value"]))
    return rows


def load_stats(path):
    """rocprofv3 --kernel-trace --stats summary (tools/profile_bench.sh): kernel -> (calls, total ns)"""
    rows = {}
    if os.path.exists(path):
        for r in csv.DictReader(open(path)):
            rows[r["Name"]] = (int(r["Calls"]), float(r["TotalDurationNs"]))
    return rows


def entry_launches(entry, table, main_pat):
    """launches of an entry in a {kernel: (calls, ...)} table; chain entries per squaring (see CHAIN_CALLS)"""
    if entry in CHAIN_CALLS and any(re.search(r"k_expo_fused_fwd2d|k_adjoint_fused2d", k) for k in table):
        return CHAIN_N * sum(c for k, (c, _) in table.items() if re.search(CHAIN_CALLS[entry], k))
    return sum(c for k, (c, _) in table.items() if re.search(main_pat, k))


def main(root, out):
    res = {"_doc": __doc__.strip().split("\n\n")[1]}
    for wl in ("cfg2", "cfg3", "cfg4", "cfg5"):
        fetch = load(os.path.join(root, "pmc_fetch_size_%s.csv" % wl))
        write = load(os.path.join(root, "pmc_write_size_%s.csv" % wl))
        if not fetch:
            continue
        res[wl] = {}
        for entry, (pats, main_pat) in ENTRIES.items():
            rb = wb = 0.0
            parts = {}
            for k, (calls, kb) in fetch.items():
                if any(re.search(p, k) for p in pats):
                    f = read_factor(k)
                    rb += calls * kb * 1024 * f
                    parts[k] = {"calls": calls, "FETCH_SIZE_KB": round(kb, 1), "read_factor": round(f, 2),
                                "WRITE_SIZE_KB": round(write.get(k, (0, 0.0))[1], 1)}
            for k, (calls, kb) in write.items():
                if any(re.search(p, k) for p in pats):
                    wb += calls * kb * 1024
            launches = entry_launches(entry, fetch, main_pat)
            if launches:
                res[wl][entry] = {"traffic_bytes_per_launch": int((rb + wb) / launches), "read_bytes_per_launch": int(rb / launches),
                                  "write_bytes_per_launch": int(wb / launches), "entry_launches_profiled": launches, "kernels": parts}
        # durations of the same entries from the kernel trace of the same command (profile_bench.sh), so that the roofline
        # fraction can be recomputed from committed files alone: algorithmic bytes / mean duration per launch / 8 TB/s
        stats = load_stats(os.path.join(root, "%s_kernel_stats.csv" % wl))
        for entry, (pats, main_pat) in ENTRIES.items():
            if entry not in res[wl] or not stats:
                continue
            ns_total = sum(t for k, (c, t) in stats.items() if any(re.search(p, k) for p in pats))
            n_l = entry_launches(entry, stats, main_pat)
            if not n_l:
                continue
            rec = res[wl][entry]
            rec["rocprof_us_per_launch"] = round(ns_total / n_l / 1e3, 2)
            rec["rocprof_launches"] = n_l
            if entry in CHAIN_CALLS and wl in PAIRED_VOXELS:
                d, nv = PAIRED_VOXELS[wl]
                alg = (8 if entry.endswith("fwd") else 12) * d * nv
                rec["algorithmic_bytes_per_launch"] = alg
                rec["frac_from_rocprof"] = round(alg / (ns_total / n_l * 1e-9) / HBM_PEAK, 4)
                rec["traffic_over_algorithmic"] = round(rec["traffic_bytes_per_launch"] / alg, 3)
    # the north-star pair on its own (tools/north_star_pair.py): every kernel of the run belongs to the pair
    ns = {}
    for level in ("init_field", "after_cfg3_ascent"):
        fetch = load(os.path.join(root, "pmc_fetch_size_ns_%s.csv" % level))
        write = load(os.path.join(root, "pmc_write_size_ns_%s.csv" % level))
        if not fetch:
            continue
        skip = re.compile(r"k_max_displacement|elementwise|distribution|fillBuffer|copyBuffer|Cijk|reduce_kernel")
        main_calls = max((c for k, (c, _) in fetch.items() if re.search(r"k_sample_march|k_sample_tiled|k_grid_sample_fwd|k_sample_ring", k)),
                         default=0)
        if not main_calls:
            continue
        rb = sum(c * kb * 1024 * read_factor(k) for k, (c, kb) in fetch.items() if not skip.search(k))
        wb = sum(c * kb * 1024 for k, (c, kb) in write.items() if not skip.search(k))
        ns[level] = {"traffic_bytes_per_launch": int((rb + wb) / main_calls), "read_bytes": int(rb / main_calls),
                     "write_bytes": int(wb / main_calls), "pairs_profiled": main_calls,
                     "kernels": {k: {"calls": c, "FETCH_SIZE_KB": round(kb, 1), "read_factor": round(read_factor(k), 2),
                                     "WRITE_SIZE_KB": round(write.get(k, (0, 0.0))[1], 1)}
                                 for k, (c, kb) in fetch.items() if not skip.search(k)}}
    if ns:
        res["north_star"] = ns
    json.dump(res, open(out, "w"), indent=1)
    for wl in res:
        if wl.startswith("_") or wl == "north_star":
            continue
        for e, v in res[wl].items():
            extra = ""
            if "rocprof_us_per_launch" in v:
                extra = "; %.1f us/launch" % v["rocprof_us_per_launch"]
            if "frac_from_rocprof" in v:
                extra += " = %.3f of peak on %.1f MB algorithmic, traffic %.2fx" % (v["frac_from_rocprof"], v["algorithmic_bytes_per_launch"] / 1e6,
                                                                                  v["traffic_over_algorithmic"])
            print("%s %-28s traffic %.1f MB/launch (read %.1f, write %.1f) over %d launches%s" % (
                wl, e, v["traffic_bytes_per_launch"] / 1e6, v["read_bytes_per_launch"] / 1e6, v["write_bytes_per_launch"] / 1e6,
                v["entry_launches_profiled"], extra))


    for level, v in res.get("north_star", {}).items():
        print("north-star pair %-18s traffic %.1f MB per fwd+bwd (read %.1f, write %.1f) against 234.9 MB algorithmic" % (
            level, v["traffic_bytes_per_launch"] / 1e6, v["read_bytes"] / 1e6, v["write_bytes"] / 1e6))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
