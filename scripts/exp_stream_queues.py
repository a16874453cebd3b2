#!/usr/bin/env python3
"""Which freshly created streams really run beside the current stream?  (HIP maps streams onto a few hardware
queues; two streams on one queue serialise.)  Probe: a spin kernel on each, wall clock of both."""
import sys, time
import torch

dev = torch.device("cuda:0")
main = torch.cuda.current_stream(dev)
N = 20_000_000


def probe(s):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    torch.cuda._sleep(N)
    with torch.cuda.stream(s):
        torch.cuda._sleep(N)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) * 1e3


torch.cuda._sleep(N)
torch.cuda.synchronize()
t0 = time.perf_counter()
torch.cuda._sleep(N)
torch.cuda.synchronize()
one = (time.perf_counter() - t0) * 1e3
print(f"one spin: {one:.2f} ms")
for prio in (0, -1):
    row = []
    for n in range(40):
        s = torch.cuda.Stream(device=dev, priority=prio)
        row.append(probe(s) / one)
    print(f"priority {prio}: " + " ".join(f"{r:.1f}" for r in row))
# the same with a non-default current stream
cur = torch.cuda.Stream(device=dev)
with torch.cuda.stream(cur):
    row = []
    for n in range(40):
        s = torch.cuda.Stream(device=dev)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        torch.cuda._sleep(N)
        with torch.cuda.stream(s):
            torch.cuda._sleep(N)
        torch.cuda.synchronize()
        row.append((time.perf_counter() - t0) * 1e3 / one)
    print("current = a pool stream, priority 0: " + " ".join(f"{r:.1f}" for r in row))
