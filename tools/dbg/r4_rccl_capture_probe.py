"""Probe (round 4): can an RCCL all-reduce be CAPTURED inside a hipGraph on this box (one rank)?  What does a replayed graph
[kernel, fork -> all-reduce on a side stream, kernel, join] cost against the same work issued eagerly?"""
import os
import socket
import time

import torch
import torch.distributed as dist

with socket.socket() as s:
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
os.environ.update(RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
dist.init_process_group("nccl", device_id=dev)
x = torch.ones(7 * 1024 * 1024, device=dev)          # 28 MB, the S-FSQ gradient buffer's size
y = torch.zeros(1024, 1024, device=dev)
dist.all_reduce(x)                                     # communicator up, eager
torch.cuda.synchronize()
side = torch.cuda.Stream()


def body():
    y.add_(1.0)
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        dist.all_reduce(x[: 3 * 1024 * 1024])
    y.mul_(1.0001)
    with torch.cuda.stream(side):
        dist.all_reduce(x[3 * 1024 * 1024:])
    torch.cuda.current_stream().wait_stream(side)
    y.add_(1.0)


st = torch.cuda.Stream()
with torch.cuda.stream(st):
    for _ in range(3):
        body()
torch.cuda.synchronize()
ok, err = True, None
g = torch.cuda.CUDAGraph()
try:
    with torch.cuda.graph(g, stream=st):
        body()
    torch.cuda.synchronize()
    for _ in range(5):
        g.replay()
    torch.cuda.synchronize()
except Exception as e:                                   # noqa: BLE001
    ok, err = False, repr(e)[:500]
print("captured RCCL all-reduce inside a hipGraph:", "works" if ok else ("FAILED " + err))
n = 200
with torch.cuda.stream(st):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        body()
    torch.cuda.synchronize()
    eager = (time.perf_counter() - t0) / n * 1e6
    if ok:
        t0 = time.perf_counter()
        for _ in range(n):
            g.replay()
        torch.cuda.synchronize()
        graph = (time.perf_counter() - t0) / n * 1e6
    else:
        graph = float("nan")
print("per iteration: eager %.1f us, graph replay %.1f us (3 small kernels + 2 one-rank all-reduces of 12 / 16 MB)" % (eager, graph))
print("x after:", float(x[0]), float(x[-1]))
dist.barrier()
dist.destroy_process_group()
