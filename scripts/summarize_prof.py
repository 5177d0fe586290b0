#!/usr/bin/env python3
"""Summarise rocprofv3 CSV output (kernel stats + PMC counter collection) for the laser_hip kernels.
usage: summarize_prof.py <dir> [kernel-substring] [--skip K]   -> markdown on stdout

Warm-only averages: when the per-dispatch kernel trace is present, every kernel also gets the average over its
dispatches in time order with the first K dropped (K = --skip, default 10 = bench.py's warm-up launches; a kernel with fewer than
3 K dispatches drops a third of them) -- the cold launches of a run (clock ramp, first-touch page mapping) are what made the
all-dispatch average sit 1.5 % above bench.py's own HIP-event time in round 4.  `roofline.frac` is reproducible from THAT column."""
import csv
import glob
import os
import sys
from collections import defaultdict

csv.field_size_limit(1 << 30)



def short(name):
    """kernel name without its parameter list; `(anonymous namespace)::` is not a parameter list (the tail kernel showed
    up as `laser_hip::`)"""
    name = name.replace("void ", "").replace("(anonymous namespace)::", "")
    depth = 0
    for i, ch in enumerate(name):
        if ch == "<":
            depth += 1
        elif ch == ">":
            depth -= 1
        elif ch == "(" and depth == 0:
            return name[:i]
    return name

def warm_table(root, pat, skip):
    for f in sorted(glob.glob(os.path.join(root, "**", "*_kernel_trace.csv"), recursive=True)):
        per = defaultdict(list)
        for r in csv.DictReader(open(f)):
            name = r.get("Kernel_Name", "")
            if pat not in name and "lh_" not in name:
                continue
            try:
                per[short(name)].append((int(r["Start_Timestamp"]), int(r["End_Timestamp"])))
            except (KeyError, ValueError):
                continue
        if not per:
            continue
        print(f"### warm-only kernel durations ({os.path.relpath(f, root)}; dispatches in time order, the first k dropped)\n")
        print("| kernel | dispatches | k dropped | warm avg ms | warm min ms | warm median ms | warm max ms | all-dispatch avg ms |")
        print("|---|---|---|---|---|---|---|---|")
        for name, spans in sorted(per.items(), key=lambda kv: -sum(e - b for b, e in kv[1])):
            spans.sort()
            d = [(e - b) / 1e6 for b, e in spans]
            k = skip if len(d) >= 3 * skip else len(d) // 3
            w = sorted(d[k:])
            print(f"| `{name}` | {len(d)} | {k} | {sum(w)/len(w):.4f} | {w[0]:.4f} | {w[len(w)//2]:.4f} | {w[-1]:.4f} | {sum(d)/len(d):.4f} |")
        print()


def main():
    args = [a for a in sys.argv[1:]]
    skip = 10
    if "--skip" in args:
        i = args.index("--skip")
        skip = int(args[i + 1])
        del args[i:i + 2]
    root = args[0]
    pat = args[1] if len(args) > 1 else "laser_hip"
    warm_table(root, pat, skip)
    for f in sorted(glob.glob(os.path.join(root, "**", "*_kernel_stats.csv"), recursive=True)):
        print(f"### kernel stats ({os.path.relpath(f, root)})\n")
        print("| kernel | calls | avg ms | min ms | max ms | % |")
        print("|---|---|---|---|---|---|")
        for r in csv.DictReader(open(f)):
            if pat in r["Name"] or "lh_" in r["Name"]:
                name = short(r["Name"])
                print(f"| `{name}` | {r['Calls']} | {float(r['AverageNs'])/1e6:.4f} | {float(r['MinNs'])/1e6:.4f} | "
                      f"{float(r['MaxNs'])/1e6:.4f} | {r['Percentage']} |")
        print()
    for f in sorted(glob.glob(os.path.join(root, "**", "*_counter_collection.csv"), recursive=True)):
        acc = defaultdict(lambda: defaultdict(list))
        dur = defaultdict(dict)
        meta = {}
        for r in csv.DictReader(open(f)):
            if pat not in r["Kernel_Name"] and "lh_" not in r["Kernel_Name"]:
                continue
            name = short(r["Kernel_Name"])
            acc[name][r["Counter_Name"]].append((r["Dispatch_Id"], float(r["Counter_Value"])))
            dur[name][r["Dispatch_Id"]] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
            meta[name] = (r["VGPR_Count"], r["Accum_VGPR_Count"], r["SGPR_Count"], r["LDS_Block_Size"], r["Scratch_Size"], r["Workgroup_Size"], r["Grid_Size"])
        for name, counters in acc.items():
            print(f"### PMC ({os.path.relpath(f, root)}) `{name}`\n")
            v = meta[name]
            d = list(dur[name].values())
            print(f"VGPR {v[0]}, AGPR {v[1]}, SGPR {v[2]}, LDS {v[3]} B, scratch {v[4]}, wg {v[5]}, grid {v[6]}; "
                  f"{len(d)} dispatches, avg {sum(d)/len(d):.4f} ms under the profiler\n")
            print("| counter | per-dispatch avg (summed over the chip) |")
            print("|---|---|")
            for c, vals in sorted(counters.items()):
                per = defaultdict(float)
                for did, val in vals:
                    per[did] += val
                avg = sum(per.values()) / len(per)
                print(f"| {c} | {avg:.6g} |")
            print()


if __name__ == "__main__":
    main()
