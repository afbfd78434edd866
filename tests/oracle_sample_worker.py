"""Child process of tests/test_shapes_gpu.py: one synthetic sample through the ORACLE's kmer-counter and seq-builder (CPU only,
never touches the GPU), results left as the reference's own files.  python oracle_sample_worker.py sample n_reads scale k b l outdir"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from metafast_amd import lib as L          # (mf_synth_reads_host is host code: the sample's definition)
from oracle import oracle

s, n, scale, k, b, l = (int(x) for x in sys.argv[1:7])
out = sys.argv[7]
bases, offsets = L.synth_reads_host(0x4D45544146415354, s, 0, n, 150, scale)
t = oracle.Table().count_buffer(bases, offsets, k)
t.write_kmers(b, os.path.join(out, f"s{s:02d}.kmers.bin"))
oracle.build_unitigs(t, k, b, l).write_fasta(os.path.join(out, f"s{s:02d}.seq.fasta"))
print(len(t))
