"""How much of a streamed count's upload window the GPU spent in kernels: rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d DIR -o p --
python3 tools/e2e_probe.py 100000000, then python3 tools/overlap_report.py DIR.  H2D copies of >= 80 us (the 8 MB sub-pieces) are clustered into windows (gaps
< 50 ms); per window: its length, the time the copy engine was busy, the time at least one kernel was running inside it."""
import csv, sys
d = sys.argv[1]
cp = list(csv.DictReader(open(d + "/p_memory_copy_trace.csv")))
kt = list(csv.DictReader(open(d + "/p_kernel_trace.csv")))
h2d = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in cp if "HOST_TO_DEVICE" in r["Direction"].upper()]
big = sorted(c for c in h2d if c[1] - c[0] >= 80_000)
ks = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("void ", "").split("<")[0]) for r in kt]
def union(ev):
    ev = sorted(ev); tot = 0
    if not ev: return 0
    a, b = ev[0]
    for s, e in ev[1:]:
        if s > b: tot += b - a; a, b = s, e
        else: b = max(b, e)
    return tot + b - a
clusters, cur = [], [big[0]]
for c in big[1:]:
    if c[0] - cur[-1][1] > 50_000_000: clusters.append(cur); cur = [c]
    else: cur.append(c)
clusters.append(cur)
for cl in clusters:
    t0, t1 = cl[0][0], max(c[1] for c in cl)
    kin = [(max(s, t0), min(e, t1), n) for s, e, n in ks if e > t0 and s < t1]
    per = {}
    for s, e, n in kin: per[n] = per.get(n, 0) + e - s
    top = ", ".join("%s %.1f" % (n, v * 1e-6) for n, v in sorted(per.items(), key=lambda kv: -kv[1])[:5])
    print("upload window %.1f ms: %d copies of >= 80 us, copy engine busy %.1f ms; kernels running for %.1f ms of it (%d launches: %s)" %
          (1e-6 * (t1 - t0), len(cl), 1e-6 * union(cl), 1e-6 * union([(s, e) for s, e, _ in kin]), len(kin), top))
