"""CPU, world_size 2 over gloo: the N > 1 path of bench.py / DESIGN_HISTORY.md section 6 -- contiguous stream sharding plus
one broadcast of the shared far-end packet per step -- gives every stream exactly the output of the unsharded
run.  The per-stream compute here is the oracle chain (no GPU in this container); on the GPU box the same
plumbing drives the HIP kernels."""
import os
import socket
import sys

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _make_inputs(n_total, n_pkts, pkt):
    from wmix_amd import synth
    far = synth.far_end(9000, n_pkts, pkt)
    near = synth.near_end(9100, n_total, n_pkts, pkt, far=far)
    return far, near


def _worker(rank, world, port, n_total, n_pkts, pkt, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle import loader
    from wmix_amd.shard import broadcast_far, stream_range
    lib = loader.port()
    far_all, near_all = _make_inputs(n_total, n_pkts, pkt)
    lo, hi = stream_range(n_total, rank, world)
    mine = near_all[lo:hi].copy()
    # only rank 0 has the far-end; everyone else receives it packet by packet
    far_recv = np.zeros_like(far_all)
    for p in range(n_pkts):
        t = torch.from_numpy(far_all[p * pkt:(p + 1) * pkt].copy()) if rank == 0 else torch.zeros(pkt, dtype=torch.int16)
        if p % 2 == 0:
            broadcast_far(t, dist, src=0)
        else:  # the overlapped form bench.py uses: start the broadcast, do other work, wait before the AEC needs it
            work = broadcast_far(t, dist, src=0, async_op=True)
            work.wait()
        far_recv[p * pkt:(p + 1) * pkt] = t.numpy()
    out = np.stack([loader.run_chain(lib, 1, 16000, 5, 15, far_recv, mine[s], pkt, prefix="orc") for s in range(hi - lo)])
    q.put((rank, lo, hi, out))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_sharding_equals_unsharded(oracle_port):
    from oracle import loader
    from wmix_amd.shard import stream_range
    n_total, n_pkts, pkt, world = 5, 60, 160, 2
    assert [stream_range(5, r, 2) for r in range(2)] == [(0, 3), (3, 5)]
    assert [stream_range(8, r, 4) for r in range(4)] == [(0, 2), (2, 4), (4, 6), (6, 8)]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_total, n_pkts, pkt, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    far, near = _make_inputs(n_total, n_pkts, pkt)
    want = np.stack([loader.run_chain(oracle_port, 1, 16000, 5, 15, far, near[s], pkt, prefix="orc") for s in range(n_total)])
    full = np.zeros_like(want)
    for rank, lo, hi, out in got:
        full[lo:hi] = out
    assert np.array_equal(full, want)
