#!/usr/bin/env python3
"""Is the host link full duplex on this box?  Pinned buffers, H2D and D2H alone and concurrently on two streams (torch)."""
import json
import time

import torch

n = 256 << 20
h_up = torch.empty(n, dtype=torch.uint8).pin_memory()
h_dn = torch.empty(n, dtype=torch.uint8).pin_memory()
d_up = torch.empty(n, dtype=torch.uint8, device="cuda")
d_dn = torch.empty(n, dtype=torch.uint8, device="cuda")
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()


def timed(fn, reps=5):
    fn(); torch.cuda.synchronize()
    best = 1e9
    for _ in range(reps):
        t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); best = min(best, time.perf_counter() - t0)
    return best


def up():
    with torch.cuda.stream(s1):
        d_up.copy_(h_up, non_blocking=True)


def down():
    with torch.cuda.stream(s2):
        h_dn.copy_(d_dn, non_blocking=True)


def both():
    up(); down()


t_up, t_dn, t_both = timed(up), timed(down), timed(both)
print(json.dumps({"bytes_each": n, "h2d_GBps": n / t_up / 1e9, "d2h_GBps": n / t_dn / 1e9, "concurrent_total_GBps": 2 * n / t_both / 1e9,
                  "concurrent_seconds": t_both, "sum_of_separate_seconds": t_up + t_dn}))
