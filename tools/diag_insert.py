import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from katome_amd import device as kd
from katome_amd.workloads import WORKLOADS
wl = WORKLOADS["c3"].scaled(int(sys.argv[1]) if len(sys.argv) > 1 else 20_000_000)
packed, skip = kd.synth_reads(0, wl.reads, wl.read_len, wl.genome_len, wl.err_rate, 0)
B = 4*1024*1024
recbuf = torch.empty(B*wl.windows_per_read, dtype=torch.int64, device="cuda")
torch.cuda.synchronize()
for it in range(2):
    b = kd.Builder(wl.k, True, table_slots_hint=int(wl.expected_distinct_canonical()*2.2))
    t00 = time.perf_counter()
    for r0 in range(0, wl.reads, B):
        nr = min(B, wl.reads-r0)
        t0 = time.perf_counter(); rec = b.extract_fixed(packed, nr, wl.read_len, None, out=recbuf, first_read=r0); t1 = time.perf_counter()
        torch.cuda.synchronize(); t2 = time.perf_counter()
        b.insert(rec); t3 = time.perf_counter()
        torch.cuda.synchronize(); t4 = time.perf_counter()
        print("it%d r0=%d extract call %.2f ms sync %.2f | insert call %.2f ms sync %.2f" % (it, r0, (t1-t0)*1e3, (t2-t1)*1e3, (t3-t2)*1e3, (t4-t3)*1e3), flush=True)
    t0 = time.perf_counter(); dg = b.finalize(); torch.cuda.synchronize(); t1 = time.perf_counter()
    print("finalize %.2f ms, total %.2f ms, edges %d" % ((t1-t0)*1e3, (t1-t00)*1e3, dg.n_edges))
    b.close()
