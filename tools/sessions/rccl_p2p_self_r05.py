"""RCCL point-to-point as Exchanger.begin / end issues it (batch_isend_irecv on slices of one send and one receive buffer, wait()), with the only peer a
one-GPU box has: the rank itself.  Checks that the API path works on this image and what a batched exchange of the size of a 1 M-element interface costs."""
import os, time
import torch
import torch.distributed as dist
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29541")
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1)
dev = torch.device("cuda", 0)
for rows in (1000, 24000, 200000):        # interface nodes x 3 components
    n = 3 * rows
    send = torch.arange(n, dtype=torch.float64, device=dev)
    recv = torch.zeros_like(send)
    h = n // 2
    def exchange():
        ops = [dist.P2POp(dist.isend, send[:h], 0), dist.P2POp(dist.irecv, recv[:h], 0), dist.P2POp(dist.isend, send[h:], 0), dist.P2POp(dist.irecv, recv[h:], 0)]
        for w in dist.batch_isend_irecv(ops):
            w.wait()
    try:
        exchange()
        torch.cuda.synchronize()
        ok = bool(torch.equal(send, recv))
        t0 = time.perf_counter()
        for _ in range(50):
            exchange()
        torch.cuda.synchronize()
        print("%7d rows: self-exchange %s, %.1f us per batched exchange (two sends + two receives)" % (rows, "correct" if ok else "WRONG", (time.perf_counter() - t0) / 50 * 1e6), flush=True)
    except Exception as e:
        print("%7d rows: %s: %s" % (rows, type(e).__name__, str(e)[:300]), flush=True)
        break
dist.destroy_process_group()
