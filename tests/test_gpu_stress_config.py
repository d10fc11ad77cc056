"""BASELINE.json configs[4] in its literal shape — 250-bp reads x 16 haplotypes per window, maxLengthDel = 10 (D = 11, the
"wider band"), haplotype lengths 120 / 160 / 200 — on the GPU: a few windows of each length against the oracle (bit-equal,
every output), and the sweep-size batch (tools/stress_sweep.py's shape, 200 reads per window) through size-independent
properties: statuses, bounds, monotone alignments, idempotence and shard invariance."""
import numpy as np
import pytest
import torch

from dindel_tgi_amd import capi, synth
from dindel_tgi_amd.device import DeviceBatch
from tests import _oracle
from tests.test_gpu_parity import assert_same

pytestmark = pytest.mark.gpu


def params():
    p = capi.params_cli_defaults()
    p.maxLengthDel = 10
    return p


@pytest.mark.parametrize("hap_len", [120, 160, 200])
def test_stress_shape_against_oracle(hap_len):
    """16 haplotypes x 24 reads of 250 bp x 3 windows per haplotype length: 1,152 pairs of ~3.5e4 cells each, every output
    equal to the oracle's (HBM-scratch builds of the D = 11 kernel: K = 2, two pairs per wavefront at K = 5 (127..158 bp), K = 4)."""
    pb = synth.generate(3, H=16, R=24, L=250, hap_len=hap_len, seed=1000 + hap_len, max_indel=3, mixed_quals=True)
    p = params()
    dev = DeviceBatch(pb, p, "cuda:0")
    dev.launch()
    torch.cuda.synchronize()
    g = capi.last_launch()
    _, G, K = next(c for c in capi.HAP_CLASSES if pb.max_hap_len <= c[0])                              # lane tiling of the longest haplotype
    assert g["D"] % 100 == 11 and g["K"] == K and g["D"] >= 100                                     # D = 11 build, back-pointers in HBM scratch
    assert capi.launch_log()[-1]["pairs_per_wave"] == G
    assert_same(dev.results(), _oracle.batch(p, pb, nthreads=16), pb)


def test_stress_shape_at_sweep_size():
    """300 windows x 16 haplotypes x 200 reads of 250 bp against 160-bp haplotypes (9.6e5 pairs, 3.9e10 cells): properties that
    do not need the oracle, plus the oracle on three windows picked from the batch."""
    pb = synth.generate(300, H=16, R=200, L=250, hap_len=160, seed=99, max_indel=3)
    p = params()
    dev = DeviceBatch(pb, p, "cuda:0")
    dev.launch()
    torch.cuda.synchronize()
    n = pb.n_pairs
    assert int((dev.out["status"][:n] != 0).sum()) == 0
    ll = dev.out["ll"][:n]
    assert bool(torch.isfinite(ll).all()) and float(ll.max()) <= 0.0
    assert bool((ll >= torch.maximum(dev.out["llOn"][:n], dev.out["llOff"][:n]) - 1e-10).all())
    # alignments: on-haplotype positions strictly increase along each read; inserted bases carry a key inside the haplotype
    hp = dev.out["hpos"][:pb.hpos_len].view(-1, 250).to(torch.int32)
    on = hp >= 0
    big = torch.where(on, hp, torch.full_like(hp, -1))
    prev = torch.cat([torch.full_like(big[:, :1], -1), torch.cummax(big, dim=1).values[:, :-1]], dim=1)
    assert bool((~on | (hp > prev)).all())
    hs = torch.from_numpy(np.repeat(np.diff(pb.a["hap_seq_off"]).astype(np.int32), 200)).to(hp.device)
    ins = hp < capi.DD_HPOS_INS_KEY0
    key = capi.DD_HPOS_INS_KEY0 - hp
    assert bool((~ins | ((key >= 1) & (key <= hs[:, None]))).all())
    fb, lb = dev.out["firstBase"][:n].to(torch.int32), dev.out["lastBase"][:n].to(torch.int32)
    assert bool(((fb >= -1) & (lb < hs) & (fb <= lb)).all())
    # idempotent: a second launch leaves every output bit-identical
    before = {k: dev.out[k].clone() for k in ("ll", "hpos", "numIndels", "offHap", "mLogBQ")}
    dev.launch()
    torch.cuda.synchronize()
    assert all(torch.equal(before[k], dev.out[k]) for k in before)
    # shard invariance + oracle: three windows computed on their own equal their slice of the batch and the oracle
    for w in (0, 137, 299):
        sh = pb.slice_windows(w, w + 1)
        d2 = DeviceBatch(sh, p, "cuda:0")
        d2.launch()
        torch.cuda.synchronize()
        p0 = int(pb.win_pair_off[w])
        assert torch.equal(d2.out["ll"][:sh.n_pairs], dev.out["ll"][p0:p0 + sh.n_pairs])
        h0 = int(pb.win_hpos_off[w])
        assert torch.equal(d2.out["hpos"][:sh.hpos_len], dev.out["hpos"][h0:h0 + sh.hpos_len])
        assert_same(d2.results(), _oracle.batch(p, sh, nthreads=16), sh)
