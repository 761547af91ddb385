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
# ---- where the host time of an exchange goes (no device synchronisation inside the loop: the enqueue cost the sequencing thread pays)
dist.init_process_group("nccl", rank=0, world_size=1)
n = 3 * 24000
send = torch.arange(n, dtype=torch.float64, device=dev); recv = torch.zeros_like(send); h = n // 2
def mk():
    return [dist.P2POp(dist.isend, send[:h], 0), dist.P2POp(dist.irecv, recv[:h], 0), dist.P2POp(dist.isend, send[h:], 0), dist.P2POp(dist.irecv, recv[h:], 0)]
for w in dist.batch_isend_irecv(mk()):
    w.wait()
torch.cuda.synchronize()
N = 200
t0 = time.perf_counter()
for _ in range(N):
    ops = mk()
t1 = time.perf_counter()
ops = mk()
works = []
for _ in range(N):
    works.append(dist.batch_isend_irecv(ops))
t2 = time.perf_counter()
for ws in works:
    for w in ws:
        w.wait()
t3 = time.perf_counter()
torch.cuda.synchronize()
t4 = time.perf_counter()
print("host time per exchange: building 4 P2POps %.1f us, batch_isend_irecv %.1f us, the waits %.1f us; device drained %.1f us later per exchange" % (
    (t1 - t0) / N * 1e6, (t2 - t1) / N * 1e6, (t3 - t2) / N * 1e6, (t4 - t3) / N * 1e6), flush=True)
# the same payload with the collective a 2-rank exchange could use instead: all_to_all_single (one call, no op list)
out = torch.empty_like(send)
dist.all_to_all_single(out, send)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(N):
    dist.all_to_all_single(out, send)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print("all_to_all_single of the same buffer: %.1f us of host time per call, %.1f us per call with the device drained" % ((t1 - t0) / N * 1e6, (t2 - t0) / N * 1e6), flush=True)
dist.destroy_process_group()
