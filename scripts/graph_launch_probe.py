#!/usr/bin/env python3
"""What does a dependent chain of tiny kernels cost per kernel on this part, enqueued one by one and replayed as a hipGraph?"""
import time, torch
D = torch.device('cuda:0')
x = torch.zeros(64, device=D)
n = 2000
def chain():
    for _ in range(n):
        x.add_(1.0)
for _ in range(2):
    chain()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
t0 = time.perf_counter(); e0.record(); chain(); e1.record(); th = time.perf_counter() - t0
torch.cuda.synchronize()
print('eager : %.2f us per kernel on the GPU timeline, host enqueue %.2f us per kernel' % (e0.elapsed_time(e1) * 1e3 / n, th * 1e6 / n))
g = torch.cuda.CUDAGraph()
s = torch.cuda.Stream()
with torch.cuda.stream(s):
    chain()
    torch.cuda.synchronize()
    with torch.cuda.graph(g, stream=s):
        chain()
torch.cuda.synchronize()
for _ in range(2):
    g.replay()
torch.cuda.synchronize()
t0 = time.perf_counter(); e0.record(); g.replay(); e1.record(); th = time.perf_counter() - t0
torch.cuda.synchronize()
print('graph : %.2f us per kernel on the GPU timeline, host %.2f us per kernel' % (e0.elapsed_time(e1) * 1e3 / n, th * 1e6 / n))
