"""A workgroup that takes SEVERAL items one after the other (persistent grids: every HBM-scratch build, and the LDS builds under the item
counter of the ragged launches) — for every lane tiling x D build x back-pointer placement the library holds.

Round 4's last fuzz campaign found what the other tests could not see: with few reads per workgroup and more items than the grid holds, a
workgroup's second and later items came out wrong in two of ~110 builds (K = 11 / D = 6 and K = 7 / D = 12 with scratch back-pointers).  Cause:
hipcc (ROCm 7.2) had put register-allocator spill code in front of the exec restore of a join block, so the lanes that sat the branch out lost a
value that lives across the item loop (tools/check_exec_spills.py finds the pattern in the assembly; hmm_kernel.hip's per-item constant loads no
longer branch).  This is the run-time side of that check: small batches whose split leaves one read per wavefront per item, five launch modes."""
import os

import numpy as np
import pytest

from dindel_tgi_amd import capi, synth
from tests import _oracle
from tests.test_gpu_parity import assert_same, run_host_api

pytestmark = pytest.mark.gpu

MODES = ({}, {"DD_DYNAMIC": "1", "DD_FORCE_GBT": "0"}, {"DD_FORCE_GBT": "1"}, {"DD_NO_HALF": "1"},
         {"DD_NO_HALF": "1", "DD_DYNAMIC": "1", "DD_FORCE_GBT": "0"})
KEYS = ("DD_DYNAMIC", "DD_FORCE_GBT", "DD_NO_HALF")


@pytest.fixture(scope="module")
def lib():
    return capi.load()


@pytest.fixture()
def clean_env():
    saved = {k: os.environ.pop(k, None) for k in KEYS}
    yield
    for k in KEYS:
        os.environ.pop(k, None)
        if saved[k] is not None:
            os.environ[k] = saved[k]


@pytest.mark.parametrize("mld,L", [(5, 100), (10, 100), (11, 100), (15, 100), (5, 36), (5, 194), (10, 194), (11, 36)])
def test_later_items_of_a_workgroup(lib, clean_env, mld, L):
    persistent_multi_round = 0
    for c, bound in enumerate(capi.HAP_CLASS_BOUNDS):
        if mld > 11 and bound > 574:                     # the D = 32 build stops at K = 9
            continue
        if bound - 4 <= mld:
            continue
        p = capi.params_cli_defaults()
        p.maxLengthDel = mld
        pb = synth.generate(14, H=8, R=48, L=L, hap_len=bound - 4, seed=77 + c, max_indel=3, sub_rate=0.01, mixed_quals=True)
        want = _oracle.batch(p, pb, nthreads=16)
        for env in MODES:
            for k in KEYS:
                os.environ.pop(k, None)
            os.environ.update(env)
            got = run_host_api(lib, p, pb)
            for l in capi.launch_log():
                if l["n_haps"] * l["split"] > 1.5 * l["grid"]:
                    persistent_multi_round += 1
            try:
                assert_same(got, want, pb)
            except AssertionError as e:
                raise AssertionError("maxLengthDel %d, %d-bp reads, haplotypes up to %d bp, %r, launches %r: %s" % (mld, L, bound, env, capi.launch_log(), str(e)[:300]))
    assert persistent_multi_round >= 20, persistent_multi_round          # the case this test exists for did occur


def test_the_fuzz_case_that_found_it(lib, clean_env):
    """seed 4100271 of tests/fuzz_campaign.py: 5 windows x 11 haplotypes of ~385 bp x 46 reads of 194 bp at maxLengthDel 11 (K = 6 and K = 7, D = 12 build)."""
    p = capi.params_cli_defaults()
    p.maxLengthDel = 11
    for sub in (0.2, 0.001):
        pb = synth.generate(5, H=11, R=46, L=194, hap_len=385, seed=4100271, max_indel=6, sub_rate=sub, vary_read_len=False, mixed_quals=True)
        assert_same(run_host_api(lib, p, pb), _oracle.batch(p, pb, nthreads=16), pb)
