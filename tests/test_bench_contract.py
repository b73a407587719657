"""bench.py's hand-off of figures that are READ from committed profiles (CPU): `roofline.traffic` comes from the PMC summary
of the kernel, and only while the kernel's sources are what they were when the counters were collected."""
import hashlib
import json
import os


def test_roofline_traffic_is_withheld_when_the_kernel_sources_changed(tmp_path, monkeypatch):
    import bench
    repo = tmp_path
    (repo / "profiles").mkdir()
    (repo / "k").mkdir()
    src = repo / "k" / "kernel.hip"
    src.write_text("__global__ void k() {}\n")
    sha = hashlib.sha256(src.read_bytes()).hexdigest()[:16]
    (repo / "profiles" / "pmc.json").write_text(json.dumps({"hbm_bytes_per_launch": 123, "kernel_sources": {"k/kernel.hip": sha}}))
    (repo / "profiles" / "old.json").write_text(json.dumps({"hbm_bytes_per_launch": 456}))          # summary of an earlier round
    monkeypatch.setattr(bench, "REPO", str(repo))
    t, note = bench.pmc_traffic_checked("pmc.json")
    assert t == 123 and "hashes checked" in note and bench.pmc_traffic("pmc.json") == 123
    t, note = bench.pmc_traffic_checked("old.json")
    assert t == 456 and "hashes" not in note
    src.write_text("__global__ void k() { /* changed */ }\n")
    t, note = bench.pmc_traffic_checked("pmc.json")
    assert t is None and "STALE" in note and "k/kernel.hip" in note
    t, note = bench.pmc_traffic_checked("missing.json")
    assert t is None and "missing" in note


def test_committed_pmc_summaries_are_either_current_or_reported_stale():
    """The two summaries the default line reads carry the hashes of their kernels' sources: the line then either quotes
    their traffic (sources unchanged) or says why it does not - never a figure of another kernel."""
    import bench
    for f in ("pmc_skinny_gemm_f32.json", "pmc_split_gemm_w16.json"):
        assert json.load(open(os.path.join(bench.REPO, "profiles", f))).get("kernel_sources"), f
        t, note = bench.pmc_traffic_checked(f)
        assert (t and "hashes checked" in note) or (t is None and "STALE" in note), (f, note)
