#!/usr/bin/env python3
"""Summarise rocprofv3 CSV output (kernel stats + PMC counter collection) for the laser_hip kernels.
usage: summarize_prof.py <dir> [kernel-substring]   -> markdown on stdout"""
import csv
import glob
import os
import sys
from collections import defaultdict

csv.field_size_limit(1 << 30)


def main():
    root = sys.argv[1]
    pat = sys.argv[2] if len(sys.argv) > 2 else "laser_hip"
    for f in sorted(glob.glob(os.path.join(root, "**", "*_kernel_stats.csv"), recursive=True)):
        print(f"### kernel stats ({os.path.relpath(f, root)})\n")
        print("| kernel | calls | avg ms | min ms | max ms | % |")
        print("|---|---|---|---|---|---|")
        for r in csv.DictReader(open(f)):
            if pat in r["Name"] or "lh_" in r["Name"]:
                name = r["Name"].split("(")[0].replace("void ", "")
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
            name = r["Kernel_Name"].split("(")[0].replace("void ", "")
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
