"""CPU: the bookkeeping of bench.py that keeps profile-derived numbers honest (they are quoted only while the kernel
sources are the ones the profile was taken on) and sizes the CPU baseline to the cores the process may really use."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def test_sources_sha_tracks_kernel_sources(tmp_path, monkeypatch):
    sha = bench.sources_sha()
    assert len(sha) == 16 and int(sha, 16) >= 0
    assert bench.sources_sha() == sha                      # deterministic
    # an extra source file under csrc changes it
    extra = os.path.join(ROOT, "em-spec_amd", "csrc", "_sha_probe_tmp.h")
    try:
        open(extra, "w").write("// probe\n")
        assert bench.sources_sha() != sha
    finally:
        os.remove(extra)
    assert bench.sources_sha() == sha


def test_profile_is_quoted_only_when_fresh(tmp_path, monkeypatch):
    prof_dir = tmp_path / "profiles"
    prof_dir.mkdir()
    monkeypatch.setattr(bench, "ROOT", str(tmp_path))
    monkeypatch.setattr(bench, "sources_sha", lambda: "aaaaaaaaaaaaaaaa")
    assert bench.profile_for("batch64") == (None, False)
    (prof_dir / "r02_batch64.json").write_text(json.dumps({"sources_sha": "bbbbbbbbbbbbbbbb", "hbm_bytes_per_launch": 1.0}))
    d, fresh = bench.profile_for("batch64")
    assert d["hbm_bytes_per_launch"] == 1.0 and not fresh   # stale: the kernels changed since
    (prof_dir / "r03_batch64.json").write_text(json.dumps({"sources_sha": "aaaaaaaaaaaaaaaa", "hbm_bytes_per_launch": 2.0}))
    d, fresh = bench.profile_for("batch64")
    assert d["hbm_bytes_per_launch"] == 2.0 and fresh and d["file"].endswith("r03_batch64.json")   # newest round first
    assert bench.profile_for("n16384") == (None, False)


def test_committed_profiles_match_the_committed_kernels():
    """The r02 summaries under profiles/ were taken on the kernel sources in this tree (otherwise bench.py withholds them)."""
    for wl in ("batch64", "n16384"):
        d, fresh = bench.profile_for(wl)
        assert d is not None, wl
        assert fresh, f"profiles/*_{wl}.json is stale: re-run tools/profile_workload.sh + tools/profile_json.py"
        assert d["hbm_bytes_per_column"] > 0 and d["valu_insts_per_column"] > 0 and 1.0 < d["clock_ghz"] < 2.6


def test_host_cores_respects_quota_and_smt():
    use, phys, logical = bench.host_cores()
    assert 1 <= use <= phys <= logical


def test_roofline_helper():
    r = bench.roofline(1000, 6144, 1.0)
    assert abs(r["achieved"] - 1000 * 6144 / 1e-3 / 1e9) < 1e-9 and r["frac"] == r["achieved"] / 8000.0
