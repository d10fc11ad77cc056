"""The device-pointer entry point does no allocation and no synchronisation, so a caller can capture it in a
HIP graph (launch-bound small batches) and replay it; results stay bit-equal to a direct launch."""
import numpy as np
import pytest
import torch

from dindel_tgi_amd import capi, synth
from dindel_tgi_amd.device import DeviceBatch

pytestmark = pytest.mark.gpu


def test_launch_is_graph_capturable(lib):
    pb = synth.generate(3, H=4, R=50, seed=8, mixed_quals=True)
    p = capi.params_cli_defaults()
    dev = DeviceBatch(pb, p, "cuda:0")
    dev.launch()                                   # warm-up outside capture (sets kernel attributes)
    want = {k: v.copy() for k, v in dev.results().items()}
    for t in dev.out.values():
        t.zero_()
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        with torch.cuda.graph(g, stream=s):
            dev.launch(stream=s)
    torch.cuda.synchronize()
    for t in dev.out.values():
        t.zero_()
    for _ in range(3):
        g.replay()
    got = dev.results()
    for k in want:
        assert np.array_equal(got[k], want[k]), k
