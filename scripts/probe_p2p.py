"""Host and device cost of one batched point-to-point exchange through torch.distributed/RCCL, measured on ONE GPU by
sending to self (RCCL allows self send/recv inside a group).  It bounds the per-exchange overhead of the sharded
V-cycle; link bandwidth is not what this measures."""
import os
import time

import torch
import torch.distributed as dist

os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29541")
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
for rows in (8, 64):
    n = rows * 16384
    a, b, c, d = (torch.zeros(n, dtype=torch.float64, device="cuda") for _ in range(4))
    a.fill_(1.0)
    c.fill_(2.0)

    def batch():
        ops = [dist.P2POp(dist.isend, a, 0), dist.P2POp(dist.irecv, b, 0), dist.P2POp(dist.isend, c, 0), dist.P2POp(dist.irecv, d, 0)]
        for w in dist.batch_isend_irecv(ops):
            w.wait()

    try:
        for _ in range(5):
            batch()
        torch.cuda.synchronize()
        assert float(b[0]) == 1.0 and float(d[0]) == 2.0
        t0 = time.perf_counter()
        for _ in range(200):
            batch()
        host = (time.perf_counter() - t0) / 200
        torch.cuda.synchronize()
        total = (time.perf_counter() - t0) / 200
        # latency of a dependent chain: exchange then a tiny kernel, repeated, with a sync each time
        t0 = time.perf_counter()
        for _ in range(100):
            batch()
            b.add_(1.0)
            torch.cuda.synchronize()
        chain = (time.perf_counter() - t0) / 100
        print("rows %d (%.1f MB per message): host enqueue %.1f us/batch, throughput %.1f us/batch, exchange+kernel+sync %.1f us"
              % (rows, n * 8 / 1e6, host * 1e6, total * 1e6, chain * 1e6), flush=True)
    except Exception as e:
        print("self send/recv failed:", repr(e), flush=True)
        break
dist.destroy_process_group()
