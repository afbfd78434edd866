#!/usr/bin/env python3
"""bench.py -- k-mers/s counted + graphed at k=31 on synthetic 150 bp reads (BASELINE.json metric).

A step = one pass of the hot path over one synthetic sample that is already resident in HBM:
  count (mask -> super-k-mer records -> LDS-staged radix partition -> LDS hash-count) -> unitigs -> [all-gather unitigs] ->
  cutter table -> connected components -> features -> [all-gather vectors] -> Bray-Curtis.
One process per GPU, one sample (100 M reads) per GPU (weak scaling).  Rank 0 prints ONE JSON line.

  python bench.py --gpus 1 --steps 3 --warmup 1
  python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port P \
         bench.py --gpus 8 --steps 3 --warmup 1
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6300 GB/s is what a copy achieves
SEED = 0x4D45544146415354      # "METAFAST"


def _load_traffic():
    """HBM bytes per launch from the rocprofv3 --pmc passes of the SAME workload (profiles/traffic_100M.json, written by
    tools/pmc_traffic.py: (2 x FETCH_SIZE + WRITE_SIZE) KB, FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for gfx950)."""
    p = os.path.join(ROOT, "profiles", "traffic_100M.json")
    try:
        return json.load(open(p))
    except Exception:
        return {}


def _load_other_shapes():
    """the hash-count kernel's roofline fraction on the workloads that are NOT the headline one (the north star's sentence is about
    the kernel, not one sample): lines of this same bench.py run by tools/other_shapes.sh on the GPU box and committed as
    profiles/hash_count_other_shapes.json -- builder-run profiles, labelled as such in the line"""
    try:
        return json.load(open(os.path.join(ROOT, "profiles", "hash_count_other_shapes.json")))
    except Exception:
        return None


def _load_extension_k63():
    """NO-REFERENCE EXTENSION, not part of the metric: one 200 M-read sample counted + graphed at k = 63 (BASELINE config 4's k = 63 leg) =
    the line `python bench.py -k 63 --reads 200000000` printed on the GPU box, committed as profiles/extension_k63_200M.json -- a builder-run
    profile, labelled as such"""
    try:
        r = json.load(open(os.path.join(ROOT, "profiles", "extension_k63_200M.json")))
        return {"label": "NO-REFERENCE EXTENSION (the reference stops at k = 31); builder-run profile profiles/extension_k63_200M.json "
                         "(python bench.py -k 63 --reads 200000000), not measured by this run",
                "k": r["config"]["k"], "reads": r["config"]["reads_per_gpu"], "ms_per_step": r["ms_per_step"], "kmers_per_s": r["value"],
                "stage_ms_per_step": r["stage_ms_per_step"], "stats": r["stats"], "kernel_ms_per_step": {n: v["ms_per_step"] for n, v in r["kernels"].items()}}
    except Exception:
        return None


def main_wide(args, ctx, device, rank, world, use_dist=False):
    """`bench.py -k K` with 32 <= K <= 63: NO-REFERENCE EXTENSION (the reference rejects k > 31, src/tools/KmersCounterMain.java:66-73;
    BASELINE config 4's k = 63 leg).  The same step -- count with the cut inside, unitigs, cutter table, components, features, matrix -- on
    2k-bit k-mers (mf_wide.hip, mf_wgraph.hip), reads resident in HBM; a normal line whose metric string says what it is.  Several GPUs: a
    sample (or several) per rank, unitigs gathered, the cutter replicated on every rank, rows all-gathered (pipeline.run_samples_wide)."""
    from metafast_amd import pipeline as P
    n_reads, rl, k = args.reads, args.read_len, args.k
    spg = max(1, args.samples_per_gpu)
    n_bases = n_reads * rl
    bases = torch.zeros(n_bases + 64, dtype=torch.uint8, device=device)
    offsets = torch.zeros(n_reads + 1, dtype=torch.int64, device=device)
    sub16k = int(round(args.sub_rate * 16384))
    ctx.synth_reads_device(SEED, rank, 0, n_reads, rl, args.genome_scale, bases.data_ptr(), offsets.data_ptr(), sub16k)
    torch.cuda.synchronize()
    gen_s = [0.0]

    def samples():
        """(as in the k <= 31 run: the reads of ONE sample in HBM at a time; the generator's time is taken out of the clock again)"""
        for j in range(spg):
            if spg > 1:
                torch.cuda.synchronize()
                g0 = time.perf_counter()
                ctx.synth_reads_device(SEED, rank * spg + j, 0, n_reads, rl, args.genome_scale, bases.data_ptr(), offsets.data_ptr(), sub16k)
                torch.cuda.synchronize()
                gen_s[0] += time.perf_counter() - g0
            yield bases, offsets, n_reads, n_bases

    def step(timings=None):
        r = P.run_samples_wide(ctx, samples(), k=k, b=args.bad_freq, l=args.min_len, b1=args.b1, b2=args.b2, device=device, timings=timings)
        stats = dict(n_occ=r["n_occ"], n_distinct=int(sum(r["n_distinct"])), n_good=int(sum(g.stats()[0] for g in r["goods"])),
                     n_unitigs=int(sum(len(q) for q in r["seqss"])), n_cutter=int(r["cutter"].stats()[0]), n_components=len(r["comps"]),
                     n_component_kmers=int(r["comps"].stats()[1]), n_reads=n_reads * spg, n_bases=n_bases * spg, n_samples=spg,
                     matrix_checksum=float(np.nansum(r["matrix"])))
        for x in r["goods"] + r["seqss"] + [r["cutter"], r["comps"]]:
            x.close()
        comm_kind[0] = (r.get("comm") or {}).get("kind")
        return stats

    comm_kind = [None]

    def barrier():
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        stats = step()
    ctx.reset_timers()
    stage_t = {}
    barrier()
    gen_s[0] = 0.0
    t0 = time.perf_counter()
    for _ in range(args.steps):
        stats = step(stage_t)
    barrier()
    elapsed = time.perf_counter() - t0 - gen_s[0]
    if use_dist:
        t = torch.tensor([elapsed], dtype=torch.float64, device=device)
        occ = torch.tensor([float(stats["n_occ"])], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dist.all_reduce(occ, op=dist.ReduceOp.SUM)
        elapsed = float(t.item())
        stats["n_occ_all_ranks"] = int(occ.item())
    if gen_s[0] and "count" in stage_t:
        stage_t["count"] -= gen_s[0]
    rep = ctx.kernel_report()
    kern = {name: dict(launches=n, ms_per_step=round(ms / max(args.steps, 1), 4), max_launch_ms=round(mx, 4)) for name, (n, ms, mx) in rep.items()}
    # the dominant kernels move, per occurrence: k_wide_kmers 16 B written (two words), the sort 4 passes x 32 B, finish / big 32 B; priced on
    # the bytes the passes must move at least once (16 B written + 16 B read per occurrence and sort pass) -- an HBM-bound integer path
    dom = max(kern, key=lambda kn: kern[kn]["ms_per_step"]) if kern else None
    roof = None
    if dom:
        passes = {"k_wide_sort": 4 * 32 + 32, "k_wide_kmers": 16 + 2 * rl / max(rl - k + 1, 1), "k_wide_finish": 32, "k_wide_big": 32}.get(dom)
        if passes:
            gb = passes * stats["n_occ"] / 1e9
            ms = kern[dom]["ms_per_step"]
            roof = dict(kernel=dom, bound="hbm", achieved=round(gb / (ms / 1e3), 1), peak=HBM_PEAK_GBS, unit="GB/s", frac=round(gb / (ms / 1e3) / HBM_PEAK_GBS, 4),
                        launch_ms=ms, algorithmic_GB=round(gb, 3), traffic=None,
                        priced_as="%.1f B per k-mer occurrence (the two 64-bit words of every occurrence through the passes of this stage)" % passes)
    out = {
        "metric": "NO-REFERENCE EXTENSION: k-mers/s counted+graphed at k=%d, %d bp reads (the reference rejects k > 31)" % (k, rl),
        "value": round(stats.get("n_occ_all_ranks", stats["n_occ"]) * args.steps / elapsed, 1), "unit": "k-mers/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(elapsed / max(args.steps, 1) * 1e3, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "u128", "data": "synthetic",
        "config": {"workload": f"{world * spg} sample(s) x {n_reads} synthetic {rl} bp reads, k={k}, {spg} per GPU, count+unitigs+components+features "
                               f"(b={args.bad_freq} l={args.min_len} b1={args.b1} b2={args.b2})",
                   "reads_per_gpu": n_reads, "read_len": rl, "k": k, "genome_scale_bp": args.genome_scale, "substitutions_per_base": round(sub16k / 16384, 5)},
        "roofline": roof, "cpu_baseline": None, "stats": stats, "comm_kind": comm_kind[0],
        "stage_ms_per_step": {kk: round(v / max(args.steps, 1) * 1e3, 3) for kk, v in stage_t.items()},
        "kernels": kern,
    }
    return out


TRAFFIC = {}
# what limits a kernel when no counters of this workload are at hand (other shapes than the headline one)
BOUND = {"k_skm_count": "lds+valu", "k_skm_scatter": "valu", "k_skm_hist": "valu"}


def bound_from_counters(tr, launch_ms):
    """roofline.bound from the PMC passes of the same workload (profiles/traffic_100M.json, tools/refresh_profiles.sh): "hbm" when
    the measured traffic alone keeps HBM half busy, else whichever of the vector ALUs / the LDS pipe the SQ counters show busiest
    (both named when they are within a third of each other); None without counters"""
    if not isinstance(tr, dict) or "valu_busy" not in tr or launch_ms <= 0:
        return None
    hbm = tr.get("hbm_GB", 0.0) / (launch_ms / 1e3) / HBM_PEAK_GBS
    v, l = tr.get("valu_busy", 0.0), tr.get("lds_busy", 0.0)
    if hbm >= 0.5 and hbm >= max(v, l):
        return "hbm"
    if max(v, l) < 0.25 and hbm < 0.25:
        return "latency"
    if l > 0 and v > 0 and min(v, l) / max(v, l) >= 0.67:
        return "lds+valu"
    return "valu" if v >= l else "lds"


def algorithmic_bytes(kernel, s):
    """Algorithmic HBM bytes of ONE full pass of `kernel` over the sample."""
    occ, dist_, good, nb = s["n_occ"], s["n_distinct"], s["n_good"], s["n_bases"]
    rec = s.get("n_records", 0) * s.get("record_bytes", 0)      # bytes of the records the counting pass partitioned
    return {
        # super-k-mer path (k >= 20): 16-byte records of about 7 k-mers each
        # (k_skm_hist: on the sample it only sizes regions from a sixteenth of the reads; the full pass runs on the cutter's
        #  small input only -- not priced)
        "k_skm_scatter": nb + nb / 8 + rec,             # + every record written once
        "k_skm_split": 3 * rec,                         # histogram read + scatter read + write
        "k_skm_count": rec + 10 * good,                 # records read + (8 B key + 2 B count) per KEPT k-mer (count > b)
        # one-record-per-k-mer path (k < 20, option skm=0)
        "k_mask": nb / 8 * 2 + s["n_reads"] * 8,
        "k_l1_hist": nb + nb / 8,                       # ASCII bases + valid-start bitmap
        "k_l1_scatter": nb + nb / 8 + 8 * occ,          # + one 8-byte k-mer written per occurrence
        "k_split": 16 * occ + 8 * occ,                  # histogram read + scatter read + write
        "k_count": 8 * occ + 10 * dist_,                # stream read + (8 B key + 2 B count) per distinct k-mer
        "k_gather": 20 * (good if rec and s.get("record_bytes") == 16 else dist_),
        "k_ut_flags": 105 * good,
    }.get(kernel)


def _cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def _reference_java(hb, ho, k, cores):
    """the real reference, only if this box has a JVM and $METAFAST_JAR points at an upstream metafast.jar
    (neither ships with this repository); timed on the same sample written out as FASTA, file reading included."""
    import shutil, subprocess, tempfile
    jar = os.environ.get("METAFAST_JAR", "")
    if not shutil.which("java") or not jar or not os.path.exists(jar):
        return "reference Java not runnable on this box (no java on PATH and/or $METAFAST_JAR not set)"
    with tempfile.TemporaryDirectory() as td:
        fa = os.path.join(td, "sample.fa")
        with open(fa, "wb") as f:
            for i in range(len(ho) - 1):
                f.write(b">%d\n" % i + hb[int(ho[i]):int(ho[i + 1])].tobytes() + b"\n")
        n_occ = int(sum(max(0, int(ho[i + 1] - ho[i]) - k + 1) for i in range(len(ho) - 1)))
        t0 = time.perf_counter()
        r = subprocess.run(["java", "-jar", jar, "-t", "kmer-counter-many", "-k", str(k), "-i", fa, "-p", str(cores), "-w", os.path.join(td, "w")],
                           capture_output=True, text=True)
        dt = time.perf_counter() - t0
        if r.returncode != 0:
            return "reference Java failed: " + r.stderr[-200:]
        return dict(value=round(n_occ / dt, 1), unit="k-mers/s", cores=cores, seconds=round(dt, 2), note="kmer-counter-many incl. JVM start and FASTA parsing")


def cpu_baseline(bases, offsets, n_reads, rl, k, args):
    """the multi-threaded C restatement of the reference's k-mer counter on this box's host cores.
    with_reader: the reference's structure end to end on a FASTA file of the first `--cpu-sample-reads` reads of the same
    sample (file written untimed): serial reader inside the dispatcher's monitor feeding P workers in 32768-read batches
    (src/io/ReadsDispatcher.java:34-53, src/io/IOUtils.java:838-865), lock-sharded linear-probing maps, single-threaded
    dump of the entries with count > 1 (IOUtils.printKmers :45-71).  count_only: the counting loop alone on reads that
    are already parsed in memory.  `value` is the with_reader figure."""
    import tempfile
    from oracle import oracle as O          # checker only: CPU baseline leg
    cores = os.cpu_count() or 1
    m = min(args.cpu_sample_reads, n_reads)
    hb = bases[: m * rl].cpu().numpy()
    td = tempfile.mkdtemp(prefix="mf_cpu_")
    fa = os.path.join(td, "sample.fa")
    rec = np.empty((m, rl + 4), dtype=np.uint8)           # ">r\n" + bases + "\n"
    rec[:, 0], rec[:, 1], rec[:, 2], rec[:, rl + 3] = ord(">"), ord("r"), 10, 10
    rec[:, 3:rl + 3] = hb.reshape(m, rl)
    rec.tofile(fa)
    del rec
    r = O.cpu_baseline_file(fa, k, cores, 1, os.path.join(td, "sample.kmers.bin"))
    wr_s = r["load_s"] + r["dump_s"]
    for f in os.listdir(td):
        os.remove(os.path.join(td, f))
    os.rmdir(td)
    m2 = min(args.cpu_count_only_reads, n_reads)
    ho = offsets[: m2 + 1].cpu().numpy().astype(np.uint64)
    c0 = time.perf_counter()
    d2, o2 = O.cpu_baseline_count(hb[: m2 * rl], ho, k, cores)
    cdt = time.perf_counter() - c0
    # ("cores" is the contract's key; what it holds is the number of THREADS used = the box's hardware threads, os.cpu_count())
    return dict(value=round(r["n_occ"] / wr_s, 1), unit="k-mers/s", cores=cores, threads=cores, kind="port", cpu_model=_cpu_model(),
                sample=f"with_reader: k-mer counter end to end on a FASTA file of the first {m} reads of the same sample "
                       f"({r['n_occ']} k-mer occurrences, {r['distinct']} distinct, {r['written']} written): serial reader + "
                       f"{cores} counting threads {r['load_s']:.2f} s, single-threaded dump {r['dump_s']:.2f} s",
                with_reader=dict(value=round(r["n_occ"] / wr_s, 1), reads=m, load_s=round(r["load_s"], 2), dump_s=round(r["dump_s"], 2)),
                count_only=dict(value=round(o2 / cdt, 1), reads=m2, seconds=round(cdt, 2),
                                note="counting loop only, reads already parsed in memory (no reader, no dump)"),
                reference_java=_reference_java(hb[: m2 * rl], ho, k, cores))


def _write_fasta(bases_dev, m, rl, path):
    """the first m reads of a sample in HBM -> a FASTA file (">r", one line per read), in pieces of 2 M reads"""
    with open(path, "wb") as f:
        for lo in range(0, m, 2_000_000):
            hi = min(m, lo + 2_000_000)
            hb = bases_dev[lo * rl: hi * rl].cpu().numpy()
            rec = np.empty((hi - lo, rl + 4), dtype=np.uint8)       # ">r\n" + bases + "\n"
            rec[:, 0], rec[:, 1], rec[:, 2], rec[:, rl + 3] = ord(">"), ord("r"), 10, 10
            rec[:, 3:rl + 3] = hb.reshape(hi - lo, rl)
            rec.tofile(f)
    return os.path.getsize(path)


def _write_fastq(bases_dev, m, rl, path):
    """the same reads as a FASTQ file ("@r", bases, "+", a line of 'I': phred 40 at offset 33)"""
    with open(path, "wb") as f:
        for lo in range(0, m, 2_000_000):
            hi = min(m, lo + 2_000_000)
            hb = bases_dev[lo * rl: hi * rl].cpu().numpy()
            rec = np.empty((hi - lo, 2 * rl + 7), dtype=np.uint8)    # "@r\n" + bases + "\n+\n" + qualities + "\n"
            rec[:, 0], rec[:, 1], rec[:, 2] = ord("@"), ord("r"), 10
            rec[:, 3:rl + 3] = hb.reshape(hi - lo, rl)
            rec[:, rl + 3], rec[:, rl + 4], rec[:, rl + 5] = 10, ord("+"), 10
            rec[:, rl + 6:2 * rl + 6] = ord("I")
            rec[:, 2 * rl + 6] = 10
            rec.tofile(f)
    return os.path.getsize(path)


def _log_steps(log_path):
    """seconds per tool run from the time stamps of the driver's log ("dd-MMM-yy  HH:mm:ss,SSS  DEBUG  Running tool X")"""
    import datetime, re
    marks = []
    try:
        for ln in open(log_path, errors="replace"):
            m = re.match(r"(\d\d-\w+-\d\d\s+\d\d:\d\d:\d\d),(\d\d\d)\s+\w+\s+(.*)", ln)
            if not m:
                continue
            t = datetime.datetime.strptime(m.group(1), "%d-%b-%y  %H:%M:%S").timestamp() + int(m.group(2)) / 1e3
            marks.append((t, m.group(3)))
    except OSError:
        return {}
    steps, cur = {}, None
    for t, msg in marks:
        if msg.startswith("Running tool "):
            if cur:
                steps[cur[1]] = round(steps.get(cur[1], 0.0) + t - cur[0], 3)
            cur = (t, msg[len("Running tool "):].strip())
    if cur and marks:
        steps[cur[1]] = round(steps.get(cur[1], 0.0) + marks[-1][0] - cur[0], 3)
    return steps


def end_to_end_and_cli(ctx, n_reads, rl, k, args, device, sample0):
    """"end-to-end k-mers/s (including file read + H2D) is a separate line".  Two figures, never `value`:
    end_to_end -- the first --e2e-reads reads of the benchmark sample as a FASTA file (written untimed, page cache) through
                  mf_count_reads_above (read + parse + H2D + count) -> unitigs -> cutter -> components -> features -> matrix,
                  the reference's path with its reader in it (src/io/IOUtils.java:772-803, src/io/ReadsDispatcher.java:34-53);
    cli        -- `metafast.sh -k K -i a.fa b.fa -w <workDir>` on two samples of --cli-reads reads each: the drop-in a user runs,
                  every step through the reference's files (kmers/*.kmers.bin, sequences/*.seq.fasta, components.bin,
                  features/*.vec, matrices/...), process start and device set-up included; seconds per tool from the workDir's log."""
    import shutil, subprocess, tempfile
    from metafast_amd import pipeline as P
    out = {}
    td = tempfile.mkdtemp(prefix="mf_e2e_", dir=os.environ.get("MF_TMPDIR"))
    sub16k = int(round(args.sub_rate * 16384))
    try:
        # (run BEFORE the benchmark sample exists: the drop-in's child process then meets the device as a user's would -- after the
        # 100 M-read steps its first kernels waited 0.9 s for memory this process had used to be cleared for it)
        big = min(args.e2e_config2_reads, n_reads) if args.e2e_config2_reads > 0 else 0
        if big and shutil.disk_usage(td).free < big * (rl + 4) * 1.2:
            out["end_to_end_config2"] = dict(skipped="no room for a %.1f GB FASTA file under %s" % (big * (rl + 4) / 1e9, td))
            big = 0
        # ---- the drop-in command line on two samples.  FIRST: the child process then meets the device as a user's would -- behind this
        # process's 100 M-read runs its first hipMallocs wait for the driver to clear the memory those had used (35 ms per GiB)
        mc = min(args.cli_reads, n_reads)
        m = min(args.e2e_reads, n_reads)
        mm = max(mc, m)
        bases = torch.zeros(mm * rl + 64, dtype=torch.uint8, device=device)
        offs = torch.zeros(mm + 1, dtype=torch.int64, device=device)
        files = []
        for j in range(2):
            ctx.synth_reads_device(SEED, sample0 + (1 - j), 0, mm, rl, args.genome_scale, bases.data_ptr(), offs.data_ptr(), sub16k)      # (last: the first reads of the benchmark sample)
            torch.cuda.synchronize()
            fj = os.path.join(td, "sample_%s.fa" % "ba"[j])
            size2 = _write_fasta(bases, mc, rl, fj)
            files.insert(0, fj)
        wd = os.path.join(td, "wd")
        t0 = time.perf_counter()
        # (the driver's default: every device the process sees, and two contexts on a device where two libraries fit side by side -- one library's
        # files are read and written while the other one is counted; `devices` / `contexts` say what the run used)
        p = subprocess.run([os.path.join(ROOT, "metafast.sh"), "-k", str(k), "-i", *files, "-w", wd, "-v"], capture_output=True, text=True, cwd=td)
        dt = time.perf_counter() - t0
        import re
        mctx = re.search(r"libraries on (\d+) device contexts", p.stderr)
        if os.environ.get("MF_IO_TIMING"):
            print("\n".join(ln for ln in p.stderr.splitlines() if ln.startswith("[mf]")), file=sys.stderr)
        occ2 = 2 * mc * (rl - k + 1)
        if p.returncode == 0:
            out["cli"] = dict(value=round(occ2 / dt, 1), unit="k-mers/s", samples=2, reads_per_sample=mc, fasta_GB=round(2 * size2 / 1e9, 3), seconds=round(dt, 3),
                              devices=max(1, torch.cuda.device_count()), contexts=int(mctx.group(1)) if mctx else 1,
                              sharded_cutter="sharded cutter table" in p.stderr,
                              step_seconds=_log_steps(os.path.join(wd, "log")),
                              what="metafast.sh -k %d -i a.fa b.fa -w wd: matrix-builder, every step through the reference's files, process start included" % k)
        else:
            out["cli"] = dict(error=p.stderr[-400:])
        shutil.rmtree(wd, ignore_errors=True)
        for f in files:
            os.remove(f)
        # ---- end to end, one process (bases: still the first reads of the benchmark sample)
        fa = os.path.join(td, "e2e.fa")
        size = _write_fasta(bases, m, rl, fa)
        fq = os.path.join(td, "e2e.fq")
        size_q = _write_fastq(bases, m, rl, fq)
        del bases, offs

        def e2e_line(path, nreads, fsize, what):
            best, occ = None, 0
            for _ in range(2):                                # (second pass: arena warm, as in a run over many samples)
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                r = P.run_samples(ctx, [(path,)], k=k, b=args.bad_freq, l=args.min_len, b1=args.b1, b2=args.b2, device=device)
                torch.cuda.synchronize()
                dt = time.perf_counter() - t0
                occ = r["n_occ"]
                for x in r["goods"] + r["seqss"] + [r["cutter"], r["comps"]]:
                    x.close()
                best = dt if best is None else min(best, dt)
            return dict(value=round(occ / best, 1), unit="k-mers/s", reads=nreads, fasta_GB=round(fsize / 1e9, 3), seconds=round(best, 4), fasta_GBps=round(fsize / 1e9 / best, 2), what=what)
        out["end_to_end"] = e2e_line(fa, m, size, "FASTA file (page cache) -> read + parse + H2D + count + unitigs + cutter + components + features + matrix, one process, tables stay in HBM")
        os.remove(fa)
        # (the same reads as FASTQ -- what a sequencer writes: twice the bytes per base across PCIe; qualities "I", nothing dropped)
        out["end_to_end_fastq"] = e2e_line(fq, m, size_q, "as end_to_end, the same reads as a FASTQ file")
        os.remove(fq)
        if big:
            # ... and where the metric is quoted: config 2's whole sample as ONE FASTA file (15.4 GB at 100 M reads), same call
            bases = torch.zeros(big * rl + 64, dtype=torch.uint8, device=device)
            offs = torch.zeros(big + 1, dtype=torch.int64, device=device)
            ctx.synth_reads_device(SEED, sample0, 0, big, rl, args.genome_scale, bases.data_ptr(), offs.data_ptr(), sub16k)
            torch.cuda.synchronize()
            fa2 = os.path.join(td, "e2e_config2.fa")
            size_b = _write_fasta(bases, big, rl, fa2)
            del bases, offs
            out["end_to_end_config2"] = e2e_line(fa2, big, size_b, "as end_to_end, on the whole sample of BASELINE config 2 as one FASTA file")
            os.remove(fa2)
    finally:
        shutil.rmtree(td, ignore_errors=True)
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--reads", type=int, default=0, help="reads per GPU (default: BASELINE.json config 2 at one GPU = 100 M, "
                                                         "config 3 = 50 M per GPU at more than one)")
    ap.add_argument("--samples-per-gpu", type=int, default=1, help="samples a rank processes one after the other (BASELINE config 5: 4)")
    ap.add_argument("--read-len", type=int, default=150)
    ap.add_argument("-k", type=int, default=31)
    ap.add_argument("--genome-scale", "--pool-scale", dest="genome_scale", type=int, default=1_000_000,
                    help="pool genome length scale in bp (1000000: the 350 Mbp pool of BASELINE configs 2-4 = 83-fold depth at 100 M reads; "
                         "5700000: the 2 Gbp pool of config 5; 16000000: 5-fold depth at 100 M reads)")
    ap.add_argument("--sub-rate", type=float, default=0.005, help="substitution errors per base (0.005: configs 2-4; 0.01: config 5)")
    ap.add_argument("--cpu-sample-reads", type=int, default=8_000_000, help="reads of the with-reader CPU baseline (FASTA file)")
    ap.add_argument("--cpu-count-only-reads", type=int, default=4_000_000, help="reads of the parser-free CPU baseline")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--e2e-reads", type=int, default=16_000_000, help="reads of the end-to-end line (FASTA file -> matrix)")
    ap.add_argument("--e2e-config2-reads", type=int, default=100_000_000, help="reads of the end_to_end_config2 line: the whole headline sample as one FASTA file (0: skip)")
    ap.add_argument("--cli-reads", type=int, default=20_000_000, help="reads per sample of the metafast.sh line (two samples)")
    ap.add_argument("--no-end-to-end", action="store_true", help="skip the end_to_end / cli keys")
    ap.add_argument("-b", dest="bad_freq", type=int, default=1, help="maximal bad frequency (1: the reference's default; 5: the CAMI example, Example.md:18-21)")
    ap.add_argument("-l", dest="min_len", type=int, default=100, help="minimal sequence length (100: default; 1200: the CAMI example)")
    ap.add_argument("--b1", type=int, default=1000)
    ap.add_argument("--b2", type=int, default=10000)
    args = ap.parse_args()

    # stdout carries ONE line, the JSON; anything libraries print there on the way (RCCL's version banner ...) goes to stderr
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.reads <= 0:
        args.reads = 100_000_000 if world == 1 else 50_000_000
    if world != args.gpus and rank == 0:
        print(f"warning: --gpus {args.gpus} but WORLD_SIZE={world}", file=sys.stderr)
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    use_dist = world > 1 or bool(os.environ.get("MF_FORCE_DIST") and "RANK" in os.environ)
    if use_dist:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        dist.init_process_group("nccl", device_id=device)      # "nccl" is RCCL on ROCm

    from metafast_amd import lib as L
    from metafast_amd import pipeline as P

    ctx = L.Context(local_rank, stream=torch.cuda.current_stream())
    ctx.set_option("profile", 1)
    for kv in filter(None, os.environ.get("MF_OPTIONS", "").split(",")):     # A/B runs: MF_OPTIONS=part_target_long=256,...
        name, val = kv.split("=")
        ctx.set_option(name, int(val))

    n_reads, rl, k = args.reads, args.read_len, args.k
    if k >= 32:
        out = main_wide(args, ctx, device, rank, world, use_dist)
        if rank == 0:
            sys.stdout.flush()
            os.dup2(real_stdout, 1)
            print(json.dumps(out), flush=True)
            os.dup2(2, 1)
        if use_dist:
            dist.destroy_process_group()
        return
    spg = max(1, args.samples_per_gpu)
    e2e = {}
    if not args.no_end_to_end and world == 1:
        try:
            e2e = end_to_end_and_cli(ctx, n_reads, rl, k, args, device, rank * spg)
        except Exception as ex:                       # (never costs the headline line)
            e2e = {"end_to_end": {"error": repr(ex)[:300]}, "cli": {"error": "not run"}}
        ctx.reset_timers()

    # ---- synthetic sample of this rank, generated in HBM (untimed) ----
    n_bases = n_reads * rl
    bases = torch.zeros(n_bases + 64, dtype=torch.uint8, device=device)
    offsets = torch.zeros(n_reads + 1, dtype=torch.int64, device=device)
    torch.cuda.synchronize()
    spg = max(1, args.samples_per_gpu)
    gen_s = [0.0]
    sub16k = int(round(args.sub_rate * 16384))
    ctx.synth_reads_device(SEED, rank * spg, 0, n_reads, rl, args.genome_scale, bases.data_ptr(), offsets.data_ptr(), sub16k)
    torch.cuda.synchronize()

    def samples():
        """this rank's samples, the reads of ONE sample in HBM at a time (the generator refills the buffer; with one
        sample per GPU nothing is generated inside the timed region)"""
        for j in range(spg):
            if spg > 1:
                torch.cuda.synchronize()
                g0 = time.perf_counter()
                ctx.synth_reads_device(SEED, rank * spg + j, 0, n_reads, rl, args.genome_scale, bases.data_ptr(), offsets.data_ptr(), sub16k)
                torch.cuda.synchronize()
                gen_s[0] += time.perf_counter() - g0        # (taken out of the timed region again: data generation is not the path)
            yield bases, offsets, n_reads, n_bases

    def barrier():
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    def step(timings=None):
        r = P.run_samples(ctx, samples(), k=k, b=args.bad_freq, l=args.min_len, b1=args.b1, b2=args.b2, device=device, timings=timings)
        nrec, rbytes = r["goods"][0].records()
        stats = dict(n_occ=r["n_occ"], n_distinct=r["n_distinct"], n_good=sum(len(g) for g in r["goods"]), n_unitigs=sum(len(q) for q in r["seqss"]),
                     n_cutter=len(r["cutter"]), n_cutter_occ=int(r["cutter"].occurrences()), n_components=len(r["comps"]), n_reads=n_reads * spg, n_bases=n_bases * spg,
                     n_records=nrec * spg, record_bytes=rbytes, n_singletons=int(sum(int(h[1]) for h in r["hists"])))
        for x in r["goods"] + r["seqss"] + [r["cutter"], r["comps"]]:
            x.close()
        for kk, v in r["comm"].items():
            comm_acc[kk] = v if isinstance(v, str) else comm_acc.get(kk, 0) + v
        return stats, r["matrix"]

    comm_acc = {}
    for _ in range(args.warmup):
        stats, _m = step()
    comm_acc.clear()
    ctx.reset_timers()
    stage_t = {}
    barrier()
    gen_s[0] = 0.0
    t0 = time.perf_counter()
    for _ in range(args.steps):
        stats, matrix = step(stage_t)
    barrier()
    elapsed = time.perf_counter() - t0 - gen_s[0]
    if gen_s[0] and "count" in stage_t:
        stage_t["count"] -= gen_s[0]          # (several samples per GPU: the generator refills the read buffer inside the count stage's clock)
    t = torch.tensor([elapsed], dtype=torch.float64, device=device)
    occ = torch.tensor([float(stats["n_occ"])], dtype=torch.float64, device=device)
    if use_dist:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dist.all_reduce(occ, op=dist.ReduceOp.SUM)
    elapsed = float(t.item())
    total_occ = float(occ.item())

    if rank == 0:
        global TRAFFIC
        if n_reads == 100_000_000 and rl == 150 and k == 31 and args.genome_scale == 1_000_000 and args.sub_rate == 0.005 and spg == 1:      # (the workload the counters were collected on)
            TRAFFIC = _load_traffic()
        rep = ctx.kernel_report()          # name -> (launches, total ms) from HIP events on the launch stream
        kern = {}
        for name, (n, ms, mx) in rep.items():
            # per step every counting kernel runs twice: once on the sample (the large launch, `max_launch_ms`) and once
            # on the unitigs for the cutter table (tiny).  The roofline is quoted on the sample launch.
            per_step_ms = ms / max(args.steps, 1)
            ab = algorithmic_bytes(name, stats)
            kern[name] = dict(launches=n, ms_per_step=round(per_step_ms, 4), max_launch_ms=round(mx, 4))
            if ab:
                # a kernel that covers the sample in ONE launch is priced on that launch; one that works through the sample
                # in batches (k_skm_count, k_gather) on all its launches of a step, the small cutter-table
                # launches included (k_skm_count's roofline counts that launch's units too, below)
                lps = n / max(args.steps, 1)
                ms_sample = mx if lps <= 2.5 and name != "k_skm_count" else per_step_ms      # (the hash-count kernel: ALL its launches of a step, always)
                kern[name]["sample_ms"] = round(ms_sample, 4)
                kern[name]["algorithmic_GB"] = round(ab / 1e9, 4)
                kern[name]["GBps"] = round(ab / 1e9 / (ms_sample / 1e3), 1) if ms_sample > 0 else None
        cands = [kn for kn in kern if "GBps" in kern[kn]]
        dom = max(cands, key=lambda kn: kern[kn]["ms_per_step"]) if cands else None

        def roof(name):
            if not name or name not in kern or not kern[name].get("GBps"):
                return None
            # (the timers carry the names of the stages' kernels; the counters those of the functions that ran: the partition-local forms)
            tr = TRAFFIC.get(name) or TRAFFIC.get({"k_ut_flags": "k_ut_flags_part", "k_cc_adjacency": "k_cc_adjacency_part", "k_gather": "k_gather_split"}.get(name, name))
            # traffic: HBM gigabytes per launch (set) from the committed rocprofv3 --pmc passes of this same workload
            # (profiles/traffic_100M.json; null for other workloads), raw counters beside it
            r = dict(kernel=name, bound=bound_from_counters(tr, kern[name]["sample_ms"]) or BOUND.get(name, "hbm"), achieved=kern[name]["GBps"], peak=HBM_PEAK_GBS, unit="GB/s",
                     frac=round(kern[name]["GBps"] / HBM_PEAK_GBS, 4),
                     launch_ms=kern[name]["sample_ms"], algorithmic_GB=kern[name]["algorithmic_GB"],
                     traffic=(tr or {}).get("hbm_GB") if isinstance(tr, dict) else None, traffic_unit="GB", traffic_counters=tr)
            if name == "k_skm_count":
                # SURVEY.md 8(d) K3, LDS-resident form: 8 B of k-mer stream per occurrence + 12 B per distinct k-mer.  The
                # kernel reads that stream as super-k-mer records (2.3 B per occurrence) and writes only the k-mers that
                # pass the cut, so the bytes it really moves (`bytes_moved_GB`, what `traffic` measures) are a quarter
                # of the figure the survey prices the step at.
                # The launch time is that of ALL the kernel's launches of a step, so the units are all theirs too: the sample's
                # occurrences and distinct k-mers + those of the cutter table's own launch (the unitigs' k-mers, a few percent).
                cut_occ, cut_dist = stats.get("n_cutter_occ", 0), stats["n_cutter"]      # (the cutter table's launch is in the time: its units are in the bytes)
                surv = 8.0 * (stats["n_occ"] + cut_occ) + 12.0 * (stats["n_distinct"] + cut_dist)
                moved, t_s = kern[name]["algorithmic_GB"], kern[name]["sample_ms"] / 1e3
                r.update(achieved=round(surv / 1e9 / t_s, 1), frac=round(surv / 1e9 / t_s / HBM_PEAK_GBS, 4),
                         algorithmic_GB=round(surv / 1e9, 4), priced_as="SURVEY 8(d) K3, LDS-resident: 8 B/occurrence + 12 B/distinct k-mer",
                         units=dict(occurrences=int(stats["n_occ"] + cut_occ), distinct=int(stats["n_distinct"] + cut_dist),
                                    of_the_cutter_table=dict(occurrences=int(cut_occ), distinct=int(cut_dist))),
                         bytes_moved_GB=moved, achieved_on_bytes_moved=kern[name]["GBps"],
                         frac_on_bytes_moved=round(kern[name]["GBps"] / HBM_PEAK_GBS, 4))
                # What binds it: LDS atomics and instruction issue, not HBM.  LDS-operation roofline beside the HBM-priced one:
                # an insert is one ds_cmpst_rtn_b64 (probe = claim) + one ds_add_u32 (count) per occurrence; tools/lds_ops.hip
                # measured 9.1 + 4.4 ns per 64-lane instruction per CU on random slots (gpurun_out/lds_ops.txt), i.e. at full lane
                # occupancy 256 CUs x 64 / 13.5 ns occurrences per second.
                lds_peak = 256 * 64 / 13.5e-9
                r["lds_roofline"] = dict(bound="lds-atomics", achieved=round(stats["n_occ"] / t_s / 1e9, 2), peak=round(lds_peak / 1e9, 1),
                                         unit="G inserts/s", frac=round(stats["n_occ"] / t_s / lds_peak, 4),
                                         model="1 ds_cmpst_rtn_b64 (9.1 ns) + 1 ds_add_u32 (4.4 ns) per 64 occurrences per CU")
            return r

        cpu = None
        if not args.no_cpu_baseline and world == 1:       # reported at N=1 only
            cpu = cpu_baseline(bases, offsets, n_reads, rl, k, args)

        out = {
            "metric": "k-mers/s counted+graphed at k=31, 150 bp reads",
            "value": round(total_occ * args.steps / elapsed, 1),
            "unit": "k-mers/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / max(args.steps, 1) * 1e3, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "u64", "data": "synthetic",
            "config": {"workload": f"{world * spg} sample(s) x {n_reads} synthetic {rl} bp reads, k={k}, {spg} sample(s) per GPU, "
                                   f"count+unitigs+components+features (b={args.bad_freq} l={args.min_len} b1={args.b1} b2={args.b2})",
                       "reads_per_gpu": n_reads, "read_len": rl, "k": k, "genome_scale_bp": args.genome_scale,
                       "substitutions_per_base": round(sub16k / 16384, 5)},
            "roofline": roof(dom),
            "roofline_hash_count": (lambda r: (r.update(other_shapes=_load_other_shapes()) or r) if isinstance(r, dict) and world == 1 else r)(
                roof("k_skm_count" if "k_skm_count" in kern else "k_count")),
            "cpu_baseline": cpu,
            "end_to_end": e2e.get("end_to_end"),
            "end_to_end_fastq": e2e.get("end_to_end_fastq"),
            "end_to_end_config2": e2e.get("end_to_end_config2"),
            "cli": e2e.get("cli"),
            "stats": stats,
            "extension_k63": _load_extension_k63() if world == 1 else None,
            # counting runs that threw their slices away and started over with more because a buffer found no place in the arena (8 x 200 M reads
            # at k = 21 on one GPU did in round 4 until the temporary lists were halved): 0 on every committed shape
            "slice_restarts": ctx.stat("slice_restarts"),
            # the sharded cutter's exchanges on rank 0 (world > 1): collectives per step, bytes received per step, seconds inside
            # them (each timed with a stream synchronisation on both sides; not measurable on this pool's 1-GPU boxes)
            "comm": {"kind": comm_acc.get("kind"),      # the library's communicator (mf_comm): "rccl" under the launcher, "local" in a single process
                     "collectives_per_step": round(comm_acc.get("collectives", 0) / max(args.steps, 1), 1),
                     "MB_received_per_step": round(comm_acc.get("bytes_in", 0) / max(args.steps, 1) / 1e6, 2),
                     "comm_ms_per_step": round(comm_acc.get("seconds", 0.0) / max(args.steps, 1) * 1e3, 3)},
            "stage_ms_per_step": {kk: round(v / max(args.steps, 1) * 1e3, 3) for kk, v in stage_t.items()},
            "kernels": kern,
        }
        sys.stdout.flush()
        os.dup2(real_stdout, 1)
        print(json.dumps(out), flush=True)
        os.dup2(2, 1)
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
