"""The exchange layer of the sharded build (katome_amd/csrc/comm.cpp) at world size 2 and 3 on CPU: ranks are processes
joined by torch.distributed/gloo, the library's communicator is given the caller's transport (katome_comm_create_callbacks,
katome_amd.shard.Comm.over_torch) and moves HOST buffers.  Under test: the variable all-to-all (counts exchanged first,
offsets, messages cut into rounds when they exceed the per-message limit), the reductions the build's control plane uses,
and how reads are split over the ranks.  No GPU call is made (the device side of the same route: tests/test_gpu_dist.py)."""
import os
import socket
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _counts(src, dst, world):
    return (7 * src + 5 * dst + 3) % 23 if (src + dst) % (world + 1) else 0          # some pairs exchange nothing


def _worker(rank, world, port, max_bytes, out_dir):
    sys.path.insert(0, ROOT)
    from katome_amd import shard as ks
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
    try:
        comm = ks.Comm.over_torch()
        assert (comm.rank, comm.world, comm.kind) == (rank, world, "callbacks")
        if max_bytes:
            comm.set_max_message_bytes(max_bytes)          # e.g. 64 bytes: 4 records of 2 words per round
        for nw in (1, 2, 3):
            counts = [_counts(rank, p, world) for p in range(world)]
            send = torch.cat([torch.arange(c * nw, dtype=torch.int64) + 1000 * p + 100000 * rank + 7 * nw for p, c in enumerate(counts)]
                             + [torch.zeros(0, dtype=torch.int64)])
            recv, rcounts = comm.exchange_host(send, counts, nw)
            want_counts = [_counts(src, rank, world) for src in range(world)]
            assert rcounts == want_counts
            want = torch.cat([torch.arange(c * nw, dtype=torch.int64) + 1000 * rank + 100000 * src + 7 * nw for src, c in enumerate(want_counts)]
                             + [torch.zeros(0, dtype=torch.int64)])
            assert torch.equal(recv, want)
        # the control plane: sums, maxima, minima of u64 vectors (counts, the 2^16-bucket histogram of global_rank)
        assert comm.allreduce([rank + 1, 10 * rank], "sum") == [world * (world + 1) // 2, 10 * world * (world - 1) // 2]
        assert comm.allreduce([rank, 5], "max") == [world - 1, 5]
        assert comm.allreduce([rank + 3], "min") == [3]
        big = comm.allreduce([(rank + 1) * (i % 97) for i in range(1 << 16)], "sum")
        assert big[:100] == [world * (world + 1) // 2 * (i % 97) for i in range(100)] and len(big) == 1 << 16
        comm.close()
        open(os.path.join(out_dir, "ok%d" % rank), "w").write("ok")
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,max_bytes", [(2, 0), (2, 64), (3, 40), (3, 8)])
def test_variable_alltoall_and_reductions_over_gloo(tmp_path, world, max_bytes):
    mp.spawn(_worker, args=(world, _free_port(), max_bytes, str(tmp_path)), nprocs=world, join=True)
    assert all(os.path.exists(os.path.join(str(tmp_path), "ok%d" % r)) for r in range(world))


def test_shard_ranges_cover_the_reads_once():
    sys.path.insert(0, ROOT)
    from katome_amd import shard as ks
    for total in (0, 1, 63, 64, 65, 700, 9000, 200_000_000, 10**9 + 7):
        for world in (1, 2, 3, 8):
            at = 0
            for r in range(world):
                first, count = ks.shard_range(total, world, r)
                assert first == at and first % 64 == 0 or count == 0       # contiguous; starts are 16-byte aligned packed rows
                at = first + count if count else at
            assert at == total
            sizes = [ks.shard_range(total, world, r)[1] for r in range(world)]
            assert max(sizes) - min(s for s in sizes if s or True) <= max(sizes)   # (trailing ranks may be empty)
            assert max(sizes) <= (total + world - 1) // world + 63
