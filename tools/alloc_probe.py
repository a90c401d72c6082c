import time, torch
torch.cuda.init(); torch.cuda.synchronize()
for gb in (8, 32, 64):
    t = time.time(); x = torch.empty(gb << 30, dtype=torch.uint8, device='cuda'); torch.cuda.synchronize(); t1 = time.time() - t
    t = time.time(); x.fill_(1); torch.cuda.synchronize(); t2 = time.time() - t
    t = time.time(); x.fill_(2); torch.cuda.synchronize(); t3 = time.time() - t
    del x; t = time.time(); torch.cuda.empty_cache(); torch.cuda.synchronize(); t4 = time.time() - t
    print("%d GiB: malloc %.3f s, first fill %.3f s, second fill %.3f s, free %.3f s" % (gb, t1, t2, t3, t4), flush=True)
