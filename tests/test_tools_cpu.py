"""Host-side measurement helpers that decide what a bench line may quote (no GPU needed)."""
import json

import pytest
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


def test_exec_spill_checker(tmp_path):
    """tools/check_exec_spills.py: the pattern as hipcc produced it in dd_hmm_kernel<11, 6, GBT> is found, the ordinary spill of a value defined in
    the branch body is not; and the assembly of the library that was built here (csrc/_asm, a by-product of `make`) is free of it."""
    import glob
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import check_exec_spills as C
    bad = """_Zkernel_bad:
	s_and_saveexec_b64 s[30:31], s[2:3]
	s_cbranch_execz .LBB0_2
	ds_read_u8 v1, v0 offset:8
	v_cndmask_b32_e32 v97, -1, v2, vcc
.LBB0_2:
	v_accvgpr_write_b32 a13, v11
	s_mov_b64 s[2:3], s[14:15]
	s_or_b64 exec, exec, s[30:31]
	s_endpgm
.Lfunc_end0:
_Zkernel_ok:
	s_and_saveexec_b64 s[30:31], s[2:3]
	s_cbranch_execz .LBB1_2
	ds_read_b64 v[4:5], v0
.LBB1_2:
	s_waitcnt lgkmcnt(0)
	scratch_store_dwordx2 off, v[4:5], off offset:104
	s_or_b64 exec, exec, s[30:31]
	s_endpgm
.Lfunc_end1:
"""
    f = tmp_path / "t.s"
    f.write_text(bad)
    found = {name: C.check(body) for name, body in C.kernels(str(f))}
    assert len(found["_Zkernel_bad"]) == 1 and found["_Zkernel_bad"][0][0] == ".LBB0_2"
    assert found["_Zkernel_ok"] == []
    built = glob.glob(os.path.join(ROOT, "dindel_tgi_amd", "csrc", "_asm", "*", "*gfx950.s"))
    if not built:
        pytest.skip("no assembly by-products here (the library was built elsewhere)")
    n = 0
    for path in built:
        for name, body in C.kernels(path):
            n += 1
            assert C.check(body) == [], (path, name)
    assert n >= 100          # every instantiation of the main kernel was looked at
