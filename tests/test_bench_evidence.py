"""bench.py reads per-launch counters from the committed rocprofv3 passes (profiles/r04_pmc.json, hbm_traffic.json) and must refuse them
when they were collected from other kernel sources than the ones that are running (round 3's review, weak #8)."""
import importlib.util
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def load_bench():
    spec = importlib.util.spec_from_file_location("bench_module", os.path.join(ROOT, "bench.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def test_counters_of_other_kernel_sources_are_refused(tmp_path):
    bench = load_bench()
    key = "hw14_1920x1080_d8_n1"
    json.dump({"csrc_sha256": "a" * 64, key: {"kernel": "k", "per_launch": {}}}, open(tmp_path / "r04_pmc.json", "w"))
    json.dump({key: {"csrc_sha256": "a" * 64, "hbm_bytes_per_launch": 123}}, open(tmp_path / "hbm_traffic.json", "w"))
    pmc, traffic, why = bench.committed_evidence(str(tmp_path), key, "a" * 64)
    assert pmc is not None and traffic == 123 and why is None
    pmc, traffic, why = bench.committed_evidence(str(tmp_path), key, "b" * 64)
    assert pmc is None and traffic is None and "other kernel sources" in why and "aaaaaaaaaaaa" in why and "bbbbbbbbbbbb" in why
    # the traffic file alone may be stale too
    json.dump({key: {"csrc_sha256": "c" * 64, "hbm_bytes_per_launch": 123}}, open(tmp_path / "hbm_traffic.json", "w"))
    pmc, traffic, why = bench.committed_evidence(str(tmp_path), key, "a" * 64)
    assert pmc is not None and traffic is None and why is None
    os.remove(tmp_path / "r04_pmc.json")
    pmc, traffic, why = bench.committed_evidence(str(tmp_path), key, "a" * 64)
    assert pmc is None and traffic is None and "not usable" in why


def test_committed_counters_are_of_this_tree(pkg):
    """What is committed under profiles/ was collected from the kernel sources of this very tree (after a kernel change:
    `bash tools/final_numbers.sh r04` on the GPU box, copy its summaries into profiles/ -- bench.py refuses the old ones until then)."""
    bench = load_bench()
    pmc, traffic, why = bench.committed_evidence(os.path.join(ROOT, "profiles"), "hw14_1920x1080_d8_n1", pkg.csrc_sha256())
    assert why is None, why
    assert pmc["hw14_1920x1080_d8_n1"]["kernel"].startswith("bvh_trace_shadow<0") and traffic and traffic > 0
