"""Does a single RCCL send/recv of more than 2 GiB arrive intact?  (Round 1 saw a 3.8 GB all-to-all come back corrupted through
torch.distributed and cut messages at 1 GiB since.)  World of one rank, the library's own RCCL transport, self send/recv:
python tools/probe_big_message.py [GiB ...]"""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from katome_amd import _lib  # noqa: E402

L = _lib.lib()
ident = (C.c_uint8 * 128)()
assert L.katome_comm_unique_id(ident) == 0, _lib.last_error()
comm = C.c_void_p()
assert L.katome_comm_create_rccl(ident, 0, 1, 0, C.byref(comm)) == 0, _lib.last_error()
for gib in [float(x) for x in sys.argv[1:]] or [0.5, 1.5, 2.5, 3.8, 6.0]:
    n = int(gib * (1 << 30)) // 8
    src = torch.arange(n, dtype=torch.int64, device="cuda") * 2654435761 + 12345
    dst = torch.zeros(n, dtype=torch.int64, device="cuda")
    for limit in (1 << 30, 1 << 40):
        dst.zero_()
        assert L.katome_comm_set_max_message_bytes(comm, limit) == 0
        sc, rc = (C.c_uint64 * 1)(n), (C.c_uint64 * 1)()
        st = L.katome_comm_exchange(comm, C.c_void_p(src.data_ptr()), sc, C.c_void_p(dst.data_ptr()), n, rc, 8, 1,
                                    C.c_void_p(torch.cuda.current_stream().cuda_stream))
        torch.cuda.synchronize()
        bad = int((dst != src).sum().item()) if st == 0 else -1
        print("%.1f GiB, message limit %s: status %d, received %d, mismatching words %d" % (gib, "1 GiB" if limit == 1 << 30 else "none", st, rc[0], bad),
              flush=True)
        if bad > 0:       # WHERE are they, and what do they hold?  (round 3: is it the probe's aliasing or RCCL's?)
            wrong = (dst != src)
            idx = torch.nonzero(wrong).flatten()
            first, last = int(idx[0]), int(idx[-1])
            edges = torch.nonzero(wrong[1:] != wrong[:-1]).flatten() + 1            # where right/wrong changes
            runs = [0] + [int(e) for e in edges[:16].tolist()]
            zeros = int((dst[wrong] == 0).sum().item())
            print("      first wrong word %d (byte offset %.3f GiB), last %d; right/wrong changes at words %s%s; wrong words that are still 0 (never written): %d of %d"
                  % (first, first * 8 / 2**30, last, runs[1:], " ..." if len(edges) > 16 else "", zeros, bad), flush=True)
            # do the wrong words hold the source's data from another offset?
            j = first
            v = int(dst[j].item())
            if v != 0:
                k = (v - 12345) // 2654435761 if (v - 12345) % 2654435761 == 0 else None
                print("      dst[%d] = src[%s]" % (j, k), flush=True)
            del wrong, idx, edges
L.katome_comm_destroy(comm)
