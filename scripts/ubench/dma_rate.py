#!/usr/bin/env python3
"""LDS-DMA request throughput per CU (see dma_rate.hip).  python scripts/ubench/dma_rate.py"""
import ctypes, os, subprocess, sys
import torch
here = os.path.dirname(os.path.abspath(__file__))
so = '/tmp/dma_rate.so'
subprocess.check_call(['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-O3', '-shared', '-fPIC', '-o', so, os.path.join(here, 'dma_rate.hip')])
lib = ctypes.CDLL(so)
lib.dma_run.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_long, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
reps = 256
per_wg = reps * 48 * 1024
for nwg in (1, 32, 256):
    src = torch.randint(0, 255, (nwg * per_wg,), dtype=torch.uint8, device='cuda')
    sink = torch.zeros(4, dtype=torch.int32, device='cuda')
    for mode, name in ((0, 'dma x4 contiguous'), (1, 'dma x4 piece-swapped'), (2, 'dma x4 all out of range'), (3, 'global_load_dwordx4 + ds_write_b128'),
                       (4, 'dma dword form (256 B)')):
        st = torch.cuda.current_stream().cuda_stream
        lib.dma_run(mode, src.data_ptr(), per_wg, reps, nwg, sink.data_ptr(), st)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(3):
            lib.dma_run(mode, src.data_ptr(), per_wg, reps, nwg, sink.data_ptr(), st)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 3
        pieces = reps * 48
        by = (256 if mode == 4 else 1024) * pieces * nwg
        print('%3d WG  %-38s %.3f ms  %.0f ns per piece per CU  %.1f GB/s per CU  %.2f TB/s total' %
              (nwg, name, ms, ms * 1e6 / pieces, by / nwg / ms / 1e6, by / ms / 1e9))
    del src
