"""HBM traffic per launch from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) -> profiles/traffic_<tag>.json
usage: python tools/pmc_traffic.py gpurun_out/pmc_<tag> profiles/traffic_100M.json
FETCH_SIZE / WRITE_SIZE are in KB; on gfx950 FETCH_SIZE reports half of the bytes of wide coalesced reads
(MI355X_MICROARCH.md, HBM section), so it is doubled.  The largest dispatch of each kernel is the sample launch."""
import csv, json, os, sys
tag, out = sys.argv[1], sys.argv[2]
res = {}
for d, cname in ((tag + "_c", "FETCH_SIZE"), (tag + "_d", "WRITE_SIZE")):
    rows = list(csv.DictReader(open(os.path.join(d, "p_counter_collection.csv"))))
    best = {}
    for r in rows:
        if r["Counter_Name"] != cname:
            continue
        name = r["Kernel_Name"].split("(")[0].replace("void ", "").split("<")[0]
        v = float(r["Counter_Value"])
        if v > best.get(name, 0):
            best[name] = v
    for n, v in best.items():
        res.setdefault(n, {})[cname] = v
final = {}
for n, v in res.items():
    f, w = v.get("FETCH_SIZE", 0.0), v.get("WRITE_SIZE", 0.0)
    final[n] = {"hbm_GB": round((2 * f + w) * 1024 / 1e9, 3), "FETCH_SIZE_KB_raw": f, "WRITE_SIZE_KB": w,
                "note": "2*FETCH_SIZE+WRITE_SIZE, per sample launch"}
json.dump(final, open(out, "w"), indent=1, sort_keys=True)
print(json.dumps({k: v["hbm_GB"] for k, v in final.items()}, indent=0))
