#!/usr/bin/env python3
"""Rewrite profiles/pmc_traffic.json from a gpu_profile_bench.sh output directory (its summary.md): per-launch
FETCH_SIZE / WRITE_SIZE of bench.py's two headline kernels, stamped with the hash of the kernel sources they were
measured on (bench.py quotes `roofline.traffic` only while that hash still matches).
Run it on the GPU box right after the passes (gpu_profile_bench.sh does), so that the hash is taken from the very tree
the counters were measured on; it writes <prof_dir>/pmc_traffic.json, which is then copied to profiles/pmc_traffic.json.
usage: update_pmc_traffic.py <prof_dir> [source-label]"""
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    prof = sys.argv[1]
    label = sys.argv[2] if len(sys.argv) > 2 else os.path.join(prof, "summary.md")
    text = open(os.path.join(prof, "summary.md")).read()
    # sections: "### PMC (<file>) `<kernel>`" ... "| COUNTER | value |"
    per = {}
    for m in re.finditer(r"### PMC \(([^)]*)\) `([^`]*)`\n\n([^\n]*)\n\n((?:\|[^\n]*\n)+)", text):
        kern, meta, table = m.group(2), m.group(3), m.group(4)
        for row in re.finditer(r"\| (\w+) \| ([0-9.e+]+) \|", table):
            per.setdefault(kern, {})[row.group(1)] = float(row.group(2))
        g = re.search(r"grid (\d+); (\d+) dispatches", meta)
        if g:
            per[kern]["_grid"] = int(g.group(1))
            per[kern]["_dispatches"] = int(g.group(2))
    import bench
    # the launch plan of each headline kernel at the headline shape, read back from the library after one launch (this script runs on
    # the GPU box, right after the counter passes)
    plans = {}
    try:
        import torch
        import laser_amd
        n = bench.SIZE
        A = torch.rand((n, n), device="cuda") * 0.2 - 0.1
        B = torch.rand((n, n), device="cuda") * 0.2 - 0.1
        C = torch.zeros((n, n), device="cuda")
        for mode, name in ((0, "laser_order"), (1, "fast")):
            laser_amd.set_float_mode(mode)
            laser_amd.matmul(A, B, 1, 0, C)
            torch.cuda.synchronize()
            plans[name] = bench.headline_plan(laser_amd)
        laser_amd.set_float_mode(0)
    except Exception as e:      # no GPU here: the file is not rewritten
        raise SystemExit(f"update_pmc_traffic.py needs the GPU the counters were collected on: {e}")
    out = {"_note": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes) on `python bench.py --steps 20 --warmup 10 "
                    "--no-cpu-baseline --no-single-process` (8192^3 sgemm, headline launches only), per launch. Both counters are "
                    "reported in KiB; FETCH_SIZE is doubled (gfx950 tallies 128-B requests at 64 B for wide coalesced loads, "
                    "MI355X_MICROARCH.md HBM section); it counts L2 fabric-side requests, Infinity-Cache hits included. "
                    "Algorithmic bytes are 0.805 GB: the rest is operand panels re-fetched by the 8 XCD-private L2s.",
           "plans": plans, "kernel_source_sha16": bench.kernel_source_sha16(plans)}
    for mode, pat in (("laser_order", r"lh_f32_exact_256x128x32"), ("fast", r"lh_f32_fast_256x256x16")):
        ks = [k for k in per if re.search(pat, k) and "FETCH_SIZE" in per[k] and "WRITE_SIZE" in per[k]]
        if not ks:
            continue
        k = ks[0]
        f, w = per[k]["FETCH_SIZE"], per[k]["WRITE_SIZE"]
        out[mode] = {"bytes_per_launch": int(round((2 * f + w) * 1024)), "fetch_size_kib": f, "write_size_kib": w,
                     "kernel": k, "grid_threads": per[k].get("_grid"), "dispatches": per[k].get("_dispatches"), "source": label}
    with open(os.path.join(prof, "pmc_traffic.json"), "w") as fh:
        json.dump(out, fh, indent=1)
        fh.write("\n")
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
