"""Host-side measurement helpers that decide what a bench line may quote (no GPU needed)."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _files(tmp_path, source_id, pairs=16000000, kernel="void ddk::dd_hmm_kernel<2, 6, false, true, 0, 1>"):
    bench = {"config": {"pairs_per_gpu": 16000000},
             "roofline": {"kernel": "dd_hmm_kernel<2, 6, false, true, 0, 1>", "traffic": None, "traffic_source": "stale: ..."}}
    pmc = {"kernel": [kernel], "config": {"pairs_per_launch": pairs, "hbm_bytes_raw": 6.4e9}, "source_id": source_id}
    b, p = tmp_path / "bench.json", tmp_path / "pmc.json"
    b.write_text(json.dumps(bench))
    p.write_text(json.dumps(pmc))
    return str(b), str(p)


def test_refresh_traffic_follows_bench_rule(tmp_path):
    """tools/refresh_traffic.py fills roofline.traffic of a profile_round bench line only from a PMC summary of the same kernel,
    the same pairs per launch and the kernel sources of this tree (the rule bench.py::measured_traffic applies to profiles/)."""
    sys.path.insert(0, ROOT)
    from dindel_tgi_amd import capi
    here = capi.kernel_source_id("dd_hmm_kernel")
    tool = os.path.join(ROOT, "tools", "refresh_traffic.py")
    b, p = _files(tmp_path, here)
    r = subprocess.run([sys.executable, tool, b, p, "profiles/rXX/d_pmc.json"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    roof = json.load(open(b))["roofline"]
    assert roof["traffic"] == 6.4e9 and roof["traffic_source"].startswith("profiles/rXX/d_pmc.json")
    for bad in (dict(source_id="0123456789ab"), dict(source_id=here, pairs=3200000), dict(source_id=here, kernel="ddk::dd_faster_kernel")):
        b, p = _files(tmp_path, **bad)
        r = subprocess.run([sys.executable, tool, b, p], capture_output=True, text=True)
        assert r.returncode != 0 and "does not match" in r.stderr
        assert json.load(open(b))["roofline"]["traffic"] is None


def test_kernel_source_id_ignores_comments_only(tmp_path):
    """capi.kernel_source_id: a hash of the kernel sources without comments and blank space — what ties profiles/*_pmc.json to a tree."""
    sys.path.insert(0, ROOT)
    from dindel_tgi_amd import capi
    a = capi.kernel_source_id("dd_hmm_kernel")
    assert len(a) == 12 and a == capi.kernel_source_id("dd_hmm_kernel")
    assert a != capi.kernel_source_id("dd_faster_kernel")
    import glob
    import pytest
    ids = [json.load(open(f)).get("source_id") for f in glob.glob(os.path.join(ROOT, "profiles", "r*", "d_pmc.json"))]
    if a not in ids:
        pytest.skip("no profiles/r*/d_pmc.json was measured on these kernel sources: bench.py will report roofline.traffic as stale "
                    "until tools/profile_round.sh d3 dd_hmm_kernel is run and its d_* files are committed")
