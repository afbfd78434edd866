"""CPU-side checks of the drop-in boundary: the library loads and exports every declared symbol."""
import ctypes
import os
import re

from metafast_amd import lib as L


def test_library_exports_every_declared_symbol():
    hdr = open(L.HEADER_PATH).read()
    declared = set(re.findall(r"\b(mf_[a-z0-9_]+)\s*\(", hdr))
    assert declared, "no declarations found"
    so = ctypes.CDLL(L.LIB_PATH)
    missing = [s for s in sorted(declared) if not hasattr(so, s)]
    assert not missing, missing
    assert declared == set(L.exported_symbols())


def test_jni_shim_matches_the_abi():
    """jni/metafast_jni.cpp (compiled only where a JDK exists) calls nothing but declared entry points, with the declared
    number of arguments; every native of jni/HipBackend.java has its Java_io_HipBackend_* definition"""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    hdr = re.sub(r"/\*.*?\*/", "", open(L.HEADER_PATH).read(), flags=re.S)
    decl = {}
    for m in re.finditer(r"\b(mf_[a-z0-9_]+)\s*\(([^;{]*?)\)\s*;", hdr, flags=re.S):
        args = m.group(2).strip()
        decl[m.group(1)] = 0 if args in ("", "void") else args.count(",") + 1
    shim = open(os.path.join(root, "jni", "metafast_jni.cpp")).read()
    calls = re.findall(r"\b(mf_[a-z0-9_]+)\s*\(", shim)
    assert len(set(calls)) >= 10
    for name in set(calls):
        assert name in decl, name
    for m in re.finditer(r"\b(mf_[a-z0-9_]+)\s*\(((?:[^()]|\([^()]*\))*)\)", shim):
        inner = re.sub(r"\([^()]*\)", "", m.group(2))
        n = 0 if not inner.strip() else inner.count(",") + 1
        assert n == decl[m.group(1)], (m.group(1), n, decl[m.group(1)])
    java = open(os.path.join(root, "jni", "HipBackend.java")).read()
    natives = re.findall(r"public static native [\w\[\]]+\s+(\w+)\(", java)
    assert len(natives) >= 12
    for n in natives:
        assert "Java_io_HipBackend_%s(" % n in shim, n


def test_no_cpu_fallback_without_gpu():
    import torch
    if torch.cuda.is_available():
        return
    try:
        L.Context(0)
    except L.MetafastError as e:
        assert "no HIP device" in str(e)
    else:
        raise AssertionError("mf_ctx_create must fail without a GPU")


def test_product_does_not_import_oracle():
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for dp, _, fs in os.walk(os.path.join(root, "metafast_amd")):
        for f in fs:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                src = open(os.path.join(dp, f), errors="replace").read()
                assert "oracle" not in src.lower().replace("# oracle", ""), f


def test_synth_host_is_deterministic():
    a, ao = L.synth_reads_host(1, 0, 0, 200, 150, 4000)
    b, bo = L.synth_reads_host(1, 0, 100, 100, 150, 4000)
    assert (a[100 * 150:] == b).all() and set(bytes(a)) <= set(b"ACGT")
