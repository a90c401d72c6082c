"""How fast do device->host copies run on this box: pageable vs pinned destination, one call vs chunks (tools only)."""
import time, torch
n = 1 << 30
d = torch.empty(n, dtype=torch.uint8, device="cuda"); d.fill_(3); torch.cuda.synchronize()
for name, make in (("pageable, fresh", lambda: torch.empty(n, dtype=torch.uint8)),
                   ("pinned, fresh", lambda: torch.empty(n, dtype=torch.uint8, pin_memory=True))):
    t = time.time(); h = make(); t_alloc = time.time() - t
    t = time.time(); h.copy_(d); torch.cuda.synchronize(); t1 = time.time() - t
    t = time.time(); h.copy_(d); torch.cuda.synchronize(); t2 = time.time() - t
    print("%-16s alloc %.3f s, first copy %.3f s (%.1f GB/s), second copy %.3f s (%.1f GB/s)" % (name, t_alloc, t1, n / 1e9 / t1, t2, n / 1e9 / t2), flush=True)
