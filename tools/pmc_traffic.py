"""HBM traffic per launch from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) -> profiles/traffic_<tag>.json
usage: python tools/pmc_traffic.py gpurun_out/pmc_<tag> profiles/traffic_100M.json
FETCH_SIZE / WRITE_SIZE are in KB; on gfx950 FETCH_SIZE reports half of the bytes of wide coalesced reads
(MI355X_MICROARCH.md, HBM section), so it is doubled.  The largest dispatch of each kernel is the sample launch; kernels that
work through the sample in several launches (k_skm_count / k_gather: batches, the walks: rounds) are summed over the
step (the passes run bench.py with --steps 1 --warmup 0, so that is one step; the small cutter-table launches are in)."""
import csv, json, os, sys
tag, out = sys.argv[1], sys.argv[2]
res = {}
for d, cname in ((tag + "_c", "FETCH_SIZE"), (tag + "_d", "WRITE_SIZE")):
    rows = list(csv.DictReader(open(os.path.join(d, "p_counter_collection.csv"))))
    best, tot, cnt = {}, {}, {}
    for r in rows:
        if r["Counter_Name"] != cname:
            continue
        name = r["Kernel_Name"].split("(")[0].replace("void ", "").split("<")[0]
        v = float(r["Counter_Value"])
        best[name] = max(best.get(name, 0.0), v)
        tot[name] = tot.get(name, 0.0) + v
        cnt[name] = cnt.get(name, 0) + 1
    # two launches (small / mid partitions) = one pass; the batches of the count
    SPLIT = ("k_ut_flags_part", "k_cc_adjacency_part", "k_dcc_adjacency_part", "k_skm_count", "k_gather", "k_ut_walk1")
    for n in best:
        res.setdefault(n, {})[cname] = tot[n] if cnt[n] > 3 or n in SPLIT else best[n]
        res[n]["dispatches"] = cnt[n]
final = {}
for n, v in res.items():
    f, w = v.get("FETCH_SIZE", 0.0), v.get("WRITE_SIZE", 0.0)
    final[n] = {"hbm_GB": round((2 * f + w) * 1024 / 1e9, 3), "FETCH_SIZE_KB_raw": f, "WRITE_SIZE_KB": w,
                "note": "2*FETCH_SIZE+WRITE_SIZE, " + ("summed over the %d launches of one step" % v["dispatches"] if v.get("dispatches", 1) > 3 or n in ("k_ut_flags_part", "k_cc_adjacency_part", "k_dcc_adjacency_part", "k_skm_count", "k_gather", "k_ut_walk1") else "per sample launch")}
# optional SQ passes (tools/pmc_bench.sh <tag> <reads> a/b ...): how busy the vector ALUs and the LDS pipe were while the kernel ran.
# Summed over the chip a counter of "cycles something was active" is compared with GRBM_GUI_ACTIVE (summed over the 8 XCDs) x 32
# CUs per XCD -- the normalisation that reproduces round 2's hand-derived 51 % / 36 % for k_skm_count.
for letter in ("a", "b"):
    f = os.path.join(tag + "_" + letter, "p_counter_collection.csv")
    if not os.path.exists(f):
        continue
    sq = {}                                              # (per pass: both carry GRBM_GUI_ACTIVE)
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"].split("(")[0].replace("void ", "").split("<")[0]
        sq.setdefault(name, {}).setdefault(r["Counter_Name"], 0.0)
        sq[name][r["Counter_Name"]] += float(r["Counter_Value"])
    for n, c in sq.items():
        gui = c.get("GRBM_GUI_ACTIVE", 0.0) * 32.0
        if n in final and gui > 0:
            if "SQ_ACTIVE_INST_VALU" in c:
                final[n]["valu_busy"] = round(c["SQ_ACTIVE_INST_VALU"] / gui, 3)
            if "SQ_LDS_IDX_ACTIVE" in c:
                final[n]["lds_busy"] = round(c["SQ_LDS_IDX_ACTIVE"] / gui, 3)
            if c.get("SQ_LDS_IDX_ACTIVE"):
                final[n]["lds_bank_conflict"] = round(c.get("SQ_LDS_BANK_CONFLICT", 0.0) / c["SQ_LDS_IDX_ACTIVE"], 3)
            for k2 in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS"):
                if k2 in c:
                    final[n][k2] = c[k2]
json.dump(final, open(out, "w"), indent=1, sort_keys=True)
print(json.dumps({k: v["hbm_GB"] for k, v in final.items()}, indent=0))
