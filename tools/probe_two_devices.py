"""First contact with more than one GPU: does the library's RCCL transport move a big message between two DIFFERENT devices
intact?  (On one card a self send/recv above 1 GiB leaves the second half of the receive buffer unwritten --
profiles/r03_rccl_big_message.txt -- which is why exchanges are cut at 1 GiB; whether that also happens across xGMI is unknown.)
Two ranks, started by this script itself before anything touches a GPU:   python tools/probe_two_devices.py [GiB ...]
Each rank sends its pattern to the other through katome_comm_exchange and checks what arrived; prints per size and message limit
the mismatching words and the effective rate per direction."""
import ctypes as C
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

if "WORLD_SIZE" not in os.environ:
    from katome_amd.launch import launch_ranks
    rc, out = launch_ranks(2, [sys.executable, os.path.abspath(__file__)] + sys.argv[1:], timeout=600)
    sys.stdout.write(out or "")
    sys.exit(rc)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402
from katome_amd import _lib  # noqa: E402
from katome_amd import shard as ks  # noqa: E402

rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
if torch.cuda.device_count() < 2:
    print("this node shows %d GPU(s): nothing to probe" % torch.cuda.device_count())
    sys.exit(0)
torch.cuda.set_device(rank)
dist.init_process_group("gloo", rank=rank, world_size=world)          # (only to hand the RCCL id round)
comm = ks.Comm.rccl(rank, world, rank)
L = _lib.lib()
peer = 1 - rank
for gib in [float(x) for x in sys.argv[1:]] or [0.25, 1.5, 3.0]:
    n = int(gib * (1 << 30)) // 8
    src = torch.arange(n, dtype=torch.int64, device="cuda") * 2654435761 + 12345 + rank
    want = torch.arange(n, dtype=torch.int64, device="cuda") * 2654435761 + 12345 + peer
    dst = torch.zeros(n, dtype=torch.int64, device="cuda")
    for limit in (1 << 30, 1 << 40):
        dst.zero_()
        comm.set_max_message_bytes(limit)
        sc = (C.c_uint64 * 2)(*[n if p == peer else 0 for p in range(2)])
        rc = (C.c_uint64 * 2)()
        torch.cuda.synchronize(); dist.barrier()
        t0 = time.perf_counter()
        st = L.katome_comm_exchange(comm._h, C.c_void_p(src.data_ptr()), sc, C.c_void_p(dst.data_ptr()), n, rc, 8, 1,
                                    C.c_void_p(torch.cuda.current_stream().cuda_stream))
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        bad = int((dst != want).sum().item()) if st == 0 else -1
        print("rank %d <- rank %d: %.2f GiB, message limit %s: status %d, mismatching words %d, %.1f GB/s per direction"
              % (rank, peer, gib, "1 GiB" if limit == 1 << 30 else "none", st, bad, n * 8 / dt / 1e9), flush=True)
comm.close()
dist.destroy_process_group()
