"""The build's guard on the neighbour-lookup kernels (metafast_amd/csrc/check_resources.py): they run at a forced occupancy and hand
data between the lanes of a wave through LDS; a build of k_ut_flags_part that spilled registers to scratch hung on the GPU, so the
Makefile fails when hipcc's resource report shows scratch for one of them."""
import glob
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(os.path.dirname(HERE), "metafast_amd", "csrc")
REMARK = "[-Rpass-analysis=kernel-resource-usage]"


def _report(name, scratch):
    rows = [("Function Name", name), ("TotalSGPRs", 106), ("VGPRs", 80), ("AGPRs", 0), ("ScratchSize [bytes/lane]", scratch), ("Dynamic Stack", "False"),
            ("Occupancy [waves/SIMD]", 6), ("SGPRs Spill", 25), ("VGPRs Spill", 0), ("LDS Size [bytes/block]", 25600)]
    out = []
    for i, (k, v) in enumerate(rows):
        out.append(f"x.hip:93:1: remark: {'' if i == 0 else '    '}{k}: {v} {REMARK}")
        out.append("   93 | __global__ void k(...) {")
        out.append("      | ^")
    return out


def _run(lines, tmp_path):
    src = tmp_path / "in.err"
    src.write_text("\n".join(lines) + "\n")
    return subprocess.run([sys.executable, os.path.join(CSRC, "check_resources.py"), str(src), str(tmp_path / "out.res")], capture_output=True, text=True)


def test_scratch_in_a_lookup_kernel_fails_the_build(tmp_path):
    ok = _run(_report("_Z15k_ut_flags_partILi1ELi31EEv13mf_index_view9ut_arraysPKmj", 0) + ["x.hip:5:1: warning: something else"], tmp_path)
    assert ok.returncode == 0 and "something else" in ok.stderr and "remark" not in ok.stderr
    assert "scratch=0 occupancy=6" in (tmp_path / "out.res").read_text()
    bad = _run(_report("_Z19k_cc_adjacency_partILi1ELi31EEv13mf_index_viewPKmS2_jiPj", 12), tmp_path)
    assert bad.returncode == 1 and "12 bytes of scratch" in bad.stderr
    other = _run(_report("_Z11k_skm_splitPK15HIP_vector_typeIyLj2EEPKm", 32), tmp_path)          # (other kernels may spill)
    assert other.returncode == 0


def test_the_guard_fails_closed(tmp_path):
    """a missing ScratchSize field, or a translation unit that should hold a lookup kernel and reports none, is a failed
    check -- not a passed one"""
    rep = [ln for ln in _report("_Z15k_ut_flags_partILi1ELi31EEv13mf_index_view9ut_arraysPKmj", 0) if "ScratchSize" not in ln]
    r = _run(rep, tmp_path)
    assert r.returncode == 1 and "no 'ScratchSize [bytes/lane]' remark" in r.stderr
    src = tmp_path / "mf_unitig.err"                  # (the unit is recognised by its file name)
    src.write_text("\n".join(_report("_Z11k_ut_linksPKm", 0)) + "\n")
    r = subprocess.run([sys.executable, os.path.join(CSRC, "check_resources.py"), str(src), str(tmp_path / "o.res")], capture_output=True, text=True)
    assert r.returncode == 1 and "k_ut_flags_part" in r.stderr
    src.write_text("\n".join(_report("_Z15k_ut_flags_partILi1ELi31EEv13mf_index_view9ut_arraysPKmj", 0)) + "\n")
    r = subprocess.run([sys.executable, os.path.join(CSRC, "check_resources.py"), str(src), str(tmp_path / "o.res")], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr


def test_the_library_in_the_tree_was_built_without_scratch_in_them():
    reports = glob.glob(os.path.join(CSRC, "build", "*.res"))
    if not reports:                                   # (a tree whose library came prebuilt: nothing to look at)
        import pytest
        pytest.skip("no build/*.res beside the sources")
    seen = 0
    for path in reports:
        for line in open(path):
            if any(n in line for n in ("k_ut_flags_part", "k_cc_adjacency_part", "k_dcc_adjacency_part")):
                seen += 1
                assert " scratch=0 " in line, line
    assert seen >= 10
