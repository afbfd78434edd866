import json, sys
d = json.load(open(sys.argv[1]))
print("value %.4g k-mers/s  ms/step %.1f" % (d["value"], d["ms_per_step"]), d["stage_ms_per_step"])
print("roofline", d["roofline"]); print("hash_count", d["roofline_hash_count"])
for k, v in sorted(d["kernels"].items(), key=lambda kv: -kv[1]["ms_per_step"])[:14]:
    print("  %-16s %s" % (k, v))
