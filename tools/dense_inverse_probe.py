#!/usr/bin/env python3
"""Time of the aggregate-level dense SPD inverse (torch.linalg.cholesky + cholesky_inverse, FP64) on the GPU
for the sizes the multi-GPU runs produce (n2 = 6 * aggregates)."""
import sys
import time

import torch

dev = torch.device("cuda", 0)
for n in [int(v) for v in (sys.argv[1:] or ["6000", "24000", "36000"])]:
    g = torch.Generator(device=dev).manual_seed(1)
    B = torch.randn(n, n, dtype=torch.float64, device=dev, generator=g)
    A = B @ B.T / n + torch.eye(n, dtype=torch.float64, device=dev)
    del B
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    L = torch.linalg.cholesky(A)
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    m = n // 8  # one rank's row block of an 8-rank run, 4096 right-hand sides at a time (as in dist.DistMG._level2)
    rows = torch.zeros(m, n, dtype=torch.float64, device=dev)
    for c0 in range(0, m, 4096):
        c1 = min(m, c0 + 4096)
        E = torch.zeros(n, c1 - c0, dtype=torch.float64, device=dev)
        E[c0:c1] = torch.eye(c1 - c0, dtype=torch.float64, device=dev)
        rows[c0:c1] = torch.cholesky_solve(E, L).T
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    err = float((rows @ A - torch.eye(n, dtype=torch.float64, device=dev)[:m]).abs().max())
    print("n %6d  cholesky %.2fs  %d rows of the inverse %.2fs  |rows A - I|max %.1e" % (n, t1 - t0, m, t2 - t1, err), flush=True)
    del A, L, rows, E
