import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from katome_amd import device as kd
n = int(sys.argv[1]) if len(sys.argv) > 1 else 800_000_000
g = torch.Generator(device="cuda"); g.manual_seed(1)
keys0 = torch.randint(0, 1 << 62, (n,), dtype=torch.int64, device="cuda", generator=g)
vals0 = torch.arange(n, dtype=torch.int32, device="cuda")
for it in range(3):
    k, v = keys0.clone(), vals0.clone()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    kd.sort_keys(k, 62, 1, v)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    ok = bool((k[1:] >= k[:-1]).all())
    print("n=%d sort(key62,val32) %.1f ms  %.2f Gkeys/s sorted=%s" % (n, dt * 1e3, n / dt / 1e9, ok), flush=True)
