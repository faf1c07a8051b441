#!/usr/bin/env python3
"""Yardstick, not product: what a plain library GEMM (torch.matmul -> hipBLASLt) sustains on this part, on random and on all-zero
operands -- the practical ceiling of the matrix pipe under the power envelope, against which the conv kernels' fractions of the
2.5 PF (bf16) / 157 TF (fp32) datasheet peaks can be read."""
import torch
D = torch.device('cuda:0')
for dt, name in ((torch.bfloat16, 'bf16'), (torch.float16, 'f16'), (torch.float32, 'f32')):
    for n in (4096, 8192):
        for kind in ('random', 'zeros'):
            a = (torch.randn(n, n, device=D) if kind == 'random' else torch.zeros(n, n, device=D)).to(dt)
            b = (torch.randn(n, n, device=D) if kind == 'random' else torch.zeros(n, n, device=D)).to(dt)
            for _ in range(3):
                c = a @ b
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            reps = 20 if dt != torch.float32 else 5
            e0.record()
            for _ in range(reps):
                c = a @ b
            e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / reps
            print('%-4s %5d^3 %-7s %8.3f ms  %7.1f TFLOP/s' % (name, n, kind, ms, 2.0 * n ** 3 / ms / 1e9), flush=True)
