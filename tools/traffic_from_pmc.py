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
    "advchain_compose_self_fwd": ([r"k_compose_self_fwd<", r"k_sample_tiled<\d, 1, \d, true>"],
                                  r"k_compose_self_fwd<|k_sample_tiled<\d, 1, \d, true>"),
    "advchain_grid_sample_fwd": ([r"k_grid_sample_fwd<", r"k_sample_tiled<\d, \d, \d, false>"],
                                 r"k_grid_sample_fwd<|k_sample_tiled<\d, \d, \d, false>"),
}
WIDE = re.compile(r"k_adjoint_gather|k_sample_tiled|k_gauss_axis_v4|k_max_displacement|k_axpy|k_absmax")   # 16 B / lane


def load(path):
    rows = {}
    if os.path.exists(path):
        for r in csv.DictReader(open(path)):
            rows[r["kernel"]] = (int(r["calls"]), float(r["mean_value"]))
    return rows


def main(root, out):
    res = {"_doc": __doc__.strip().split("\n\n")[1]}
    for wl in ("cfg2", "cfg3"):
        fetch = load(os.path.join(root, "pmc_fetch_size_%s.csv" % wl))
        write = load(os.path.join(root, "pmc_write_size_%s.csv" % wl))
        if not fetch:
            continue
        res[wl] = {}
        for entry, (pats, main_pat) in ENTRIES.items():
            rb = wb = 0.0
            launches = 0
            parts = {}
            for k, (calls, kb) in fetch.items():
                if any(re.search(p, k) for p in pats):
                    f = 2.0 if WIDE.search(k) else 1.0 / 0.84
                    rb += calls * kb * 1024 * f
                    parts[k] = {"calls": calls, "FETCH_SIZE_KB": round(kb, 1), "read_factor": round(f, 2),
                                "WRITE_SIZE_KB": round(write.get(k, (0, 0.0))[1], 1)}
                    if re.search(main_pat, k):
                        launches += calls
            for k, (calls, kb) in write.items():
                if any(re.search(p, k) for p in pats):
                    wb += calls * kb * 1024
            if launches:
                res[wl][entry] = {"traffic_bytes_per_launch": int((rb + wb) / launches), "read_bytes_per_launch": int(rb / launches),
                                  "write_bytes_per_launch": int(wb / launches), "entry_launches_profiled": launches, "kernels": parts}
    json.dump(res, open(out, "w"), indent=1)
    for wl in res:
        if wl.startswith("_"):
            continue
        for e, v in res[wl].items():
            print("%s %-28s traffic %.1f MB/launch (read %.1f, write %.1f) over %d launches" % (
                wl, e, v["traffic_bytes_per_launch"] / 1e6, v["read_bytes_per_launch"] / 1e6, v["write_bytes_per_launch"] / 1e6,
                v["entry_launches_profiled"]))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
