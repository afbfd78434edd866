"""Summarise rocprofv3 --pmc passes: python tools/pmc_report.py gpurun_out/pmc_<tag>   (reads _a.._d dirs)"""
import csv, sys, collections, glob, os
tag = sys.argv[1]
data = collections.defaultdict(dict)     # kernel -> counter -> value (largest dispatch of that kernel)
dur = {}
for d in sorted(glob.glob(tag + "_*/")):
    rows = list(csv.DictReader(open(os.path.join(d, "p_counter_collection.csv"))))
    tr = {r["Dispatch_Id"]: r for r in csv.DictReader(open(os.path.join(d, "p_kernel_trace.csv")))}
    best = {}
    for r in rows:
        name = r["Kernel_Name"].split("(")[0].replace("void ", "")
        t = tr.get(r["Dispatch_Id"])
        du = (int(t["End_Timestamp"]) - int(t["Start_Timestamp"])) if t else 0
        key = (name, r["Counter_Name"])
        if key not in best or du > best[key][0]:
            best[key] = (du, float(r["Counter_Value"]))
    for (name, c), (du, v) in best.items():
        data[name][c] = v
        dur[name] = max(dur.get(name, 0), du)
names = sorted(dur, key=lambda n: -dur[n])[:12]
for n in names:
    c = data[n]
    print(f"== {n}  {dur[n]/1e6:.2f} ms")
    for k in sorted(c):
        print(f"   {k:24s} {c[k]:.4g}")
