import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from dindel_tgi_amd import capi
from dindel_tgi_amd.batch import pack, pair_slices
from tests import _oracle
from tests.test_gpu_fuzz import make_windows
from tests.test_gpu_parity import run_host_api
rng = np.random.default_rng(1011)
ws = make_windows(rng, 120, 30, 300, min_hap=3)
p = capi.params_cli_defaults(); p.maxLengthDel = 3
pb = pack(ws)
lib = capi.load()
got = run_host_api(lib, p, pb)
want = _oracle.batch(p, pb, nthreads=8)
n = pb.n_pairs
for k in ("ll", "llOn", "llOff", "offHap", "offHapHMQ", "numIndels", "firstBase", "lastBase", "status"):
    bad = np.nonzero(got[k][:n] != want[k][:n])[0]
    print(k, len(bad), bad[:10])
bad = np.nonzero(got["offHap"][:n] != want["offHap"][:n])[0]
for pidx in bad[:3]:
    w = int(np.searchsorted(pb.win_pair_off, pidx, side="right") - 1)
    p0, H, R, hp0, SL, rs0 = pair_slices(pb, w)
    h, r = divmod(pidx - p0, R)
    W = ws[w]
    print("pair", pidx, "win", w, "h", h, "r", r, "hap", W.haps[h], "read", W.reads[r].seq, "start", W.reads[r].start, "mq", W.reads[r].mapQual, "unm", W.reads[r].unmapped)
    print(" got ll %.17g want %.17g; offHap %d/%d offHapHMQ %d/%d llOn %.17g/%.17g llOff %.17g/%.17g" % (got["ll"][pidx], want["ll"][pidx], got["offHap"][pidx], want["offHap"][pidx], got["offHapHMQ"][pidx], want["offHapHMQ"][pidx], got["llOn"][pidx], want["llOn"][pidx], got["llOff"][pidx], want["llOff"][pidx]))
