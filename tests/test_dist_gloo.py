"""world_size-2 gloo test of the N>1 path: contiguous window shards per rank + gather of per-pair records to
rank 0 reproduce the single-process result in window order.  On CPU the per-shard compute stand-in is the
oracle (this is a test of sharding and the collective, not of the kernel)."""
import os
import socket
import sys

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out_path):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from dindel_tgi_amd import capi, synth
    from dindel_tgi_amd.shard import gather_records, window_block
    from tests import _oracle
    pb = synth.generate(5, H=3, R=6, L=30, hap_len=40, seed=9, vary_read_len=True, mixed_quals=True)
    p = capi.params_cli_defaults()
    w0, w1 = window_block(pb.n_windows, rank, world)
    shard = pb.slice_windows(w0, w1)
    res = _oracle.batch(p, shard)
    ll = gather_records(torch.from_numpy(res["ll"][:shard.n_pairs].copy()))
    flags = gather_records(torch.from_numpy(res["offHapHMQ"][:shard.n_pairs].copy()))
    if rank == 0:
        full = _oracle.batch(p, pb)
        ok = bool(np.array_equal(ll.numpy(), full["ll"][:pb.n_pairs]) and
                  np.array_equal(flags.numpy(), full["offHapHMQ"][:pb.n_pairs]))
        open(out_path, "w").write("ok" if ok else "mismatch")
    else:
        assert ll is None and flags is None
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_shard_and_gather(tmp_path):
    out = str(tmp_path / "result.txt")
    mp.spawn(_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    assert open(out).read() == "ok"
