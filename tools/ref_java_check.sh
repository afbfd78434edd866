#!/bin/bash
# Pins the oracle's UNPINNED rules against the real reference, where one can run it: needs `java` on PATH and
# $METAFAST_JAR = an upstream metafast.jar (neither ships with this repository; there is no JDK in the build image).
# Runs the reference and the oracle on tests/golden/ref_test_data + the branchy fixture (tests/util.py: branchy_reads,
# seeds 7 8 9) and compares sorted .kmers.bin records, strand-normalised unitig multisets and components.bin partitions.
#   tools/ref_java_check.sh [workdir]
set -e
cd "$(dirname "$0")/.."
if ! command -v java >/dev/null || [ -z "$METAFAST_JAR" ] || [ ! -f "$METAFAST_JAR" ]; then
    echo "reference Java not runnable here (java on PATH and \$METAFAST_JAR needed): nothing checked"; exit 2
fi
WD=${1:-/tmp/mf_ref_check}; rm -rf "$WD"; mkdir -p "$WD"
python3 - "$WD" <<'PY'
import os, subprocess, sys, numpy as np
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
from oracle import oracle as O
from util import branchy_reads, canon_seq
wd = sys.argv[1]
files = [os.path.abspath(f"tests/golden/ref_test_data/meta_test_{i}.fa") for i in (1, 2, 3)]
for seed in (7, 8, 9):
    b, o = branchy_reads(seed)
    f = os.path.join(wd, f"branchy_{seed}.fa")
    with open(f, "wb") as fh:
        for j in range(len(o) - 1):
            fh.write(b">r%d\n" % j + b[int(o[j]):int(o[j + 1])].tobytes() + b"\n")
    files.append(f)
jar = os.environ["METAFAST_JAR"]
def java(*a):
    subprocess.run(["java", "-jar", jar, *a], check=True, stdout=subprocess.DEVNULL)
bad = 0
for group, fs, b1, b2 in (("ref", files[:3], 1000, 10000), ("branchy", files[3:], 100, 1000)):
    w = os.path.join(wd, group)
    java("-t", "matrix-builder", "-k", "31", "-i", *fs, "-b1", str(b1), "-b2", str(b2), "-w", w)
    r = O.run_pipeline(fs, b1=b1, b2=b2)
    for f, s in zip(sorted(fs), r["samples"]):
        name = os.path.basename(f)[:-3]
        rec = np.fromfile(os.path.join(w, "kmer-counter-many", "kmers", name + ".kmers.bin"), dtype=np.uint8).reshape(-1, 10)
        jk = np.sort(rec[:, :8].copy().view(">u8").ravel().astype(np.uint64))
        ok = np.sort(s["good"].export()[0])
        if not np.array_equal(jk, ok): print("MISMATCH kmers", name); bad += 1
        js = sorted(canon_seq("".join(x.split("\n")[1:])) for x in open(os.path.join(w, "seq-builder-many", "sequences", name + ".seq.fasta")).read().split(">")[1:])
        os_ = sorted(canon_seq(q[0]) for q in s["seqs"].all())
        if js != os_: print("MISMATCH unitigs", name, len(js), len(os_)); bad += 1
    jc = O.load_components(os.path.join(w, "component-cutter", "components.bin")).all()
    part = lambda cs: sorted(tuple(sorted(int(x) for x in km)) for _, _, _, km in cs)
    if part(jc) != part(r["comps"].all()): print("MISMATCH components", group); bad += 1
    jm = np.loadtxt([l for l in open([os.path.join(w, "matrices", x) for x in os.listdir(os.path.join(w, "matrices")) if "original_order" in x][0]) if not l.startswith("#")], usecols=range(1, len(fs) + 1))
    if np.abs(jm - np.round(r["matrix"], 4)).max() > 1e-4: print("MISMATCH matrix", group); bad += 1
print("reference check:", "OK" if not bad else f"{bad} mismatch(es)")
sys.exit(1 if bad else 0)
PY
