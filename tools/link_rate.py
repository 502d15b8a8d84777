"""Host link rates seen by one process: pageable / pinned, each direction alone and both at once (two threads, two streams)."""
import threading
import time

import torch

n = 1 << 30
dev = torch.device("cuda:0")
d_a = torch.empty(n, dtype=torch.uint8, device=dev)
d_b = torch.empty(n, dtype=torch.uint8, device=dev)
for pinned in (False, True):
    h_a = torch.empty(n, dtype=torch.uint8, pin_memory=pinned)
    h_b = torch.empty(n, dtype=torch.uint8, pin_memory=pinned)
    h_a.fill_(1)
    h_b.fill_(2)
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()

    def up():
        with torch.cuda.stream(s1):
            d_a.copy_(h_a, non_blocking=True)
        s1.synchronize()

    def down():
        with torch.cuda.stream(s2):
            h_b.copy_(d_b, non_blocking=True)
        s2.synchronize()

    def timed(fs):
        best = 1e9
        for _ in range(3):
            torch.cuda.synchronize()
            t = time.perf_counter()
            th = [threading.Thread(target=f) for f in fs]
            [x.start() for x in th]
            [x.join() for x in th]
            best = min(best, time.perf_counter() - t)
        return best

    for name, fs in (("H2D", [up]), ("D2H", [down]), ("both", [up, down])):
        t = timed(fs)
        print(f"pinned={pinned} {name}: {len(fs) * n / t / 1e9:.1f} GB/s total ({t * 1e3:.1f} ms)", flush=True)
