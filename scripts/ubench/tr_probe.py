import ctypes, os, torch
L = ctypes.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'tr_probe.so'))
def run(addr):
    a = torch.tensor(addr, dtype=torch.int32, device='cuda')
    o = torch.zeros(256, dtype=torch.int16, device='cuda')
    L.tr_probe(ctypes.c_void_p(a.data_ptr()), ctypes.c_void_p(o.data_ptr()), ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
    torch.cuda.synchronize()
    return o.cpu().view(64, 4).tolist()
# experiment 1: lane l -> address 8*l (contiguous 8-byte chunks): elements 4l..4l+3
r = run([8 * l for l in range(64)])
print('contiguous chunks (lane l loads elements 4l..4l+3):')
for l in range(0, 64, 1): print(l, r[l])
# experiment 2: matrix [voxel k][channel c] with pitch 64 elements (128 B): lane (r=(l&15)>>2, c4=l&3, group g=l>>4): addr = ((r + 4*(g>>1))*64 + 16*(g&1) + 4*c4)*2
addr = []
for l in range(64):
    g, r_, c4 = l >> 4, (l & 15) >> 2, l & 3
    addr.append(((r_ + 4 * (g >> 1)) * 64 + 16 * (g & 1) + 4 * c4) * 2)
r = run(addr)
print('matrix pitch 64: expect lane i of group g to get column 16*(g&1)+ (l&15), rows 4*(g>>1)+0..3 -> element row*64+col')
for l in range(64): print(l, r[l], [ (4*(l>>5) + j)*64 + 16*((l>>4)&1) + (l&15) for j in range(4)])
