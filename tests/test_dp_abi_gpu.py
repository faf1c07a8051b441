"""-m gpu: the data-parallel exchange entry points of the C ABI (bts_dp_*: SURVEY 8b / 8e -- C1 bucketed gradient all-reduce, C2 parameter
broadcast, C3 small fp64 all-reduce over RCCL on device buffers) on a ONE-rank communicator made through the library itself: the sums
over one rank are the identity, the calls must be enqueued on the caller's stream and validate their arguments.  (Two ranks need two
devices: tests/test_dp_gpu.py::test_two_ranks_over_rccl_on_device_buffers covers the Python host's exchange there.)"""
import ctypes

import pytest
import torch

pytestmark = pytest.mark.gpu


def test_one_rank_communicator_through_the_c_abi():
    import bts_amd  # noqa: F401
    from bts_amd._lib import lib
    L = lib()
    if not L._bts_dp_available():
        pytest.skip('no RCCL library on this box')
    dev = torch.device('cuda', 0)
    torch.cuda.set_device(dev)
    nid = L._bts_dp_unique_id_bytes()
    assert nid >= 128
    uid = (ctypes.c_char * nid)()
    L.call('bts_dp_comm_unique_id', ctypes.cast(uid, ctypes.c_void_p))
    comm = ctypes.c_void_p()
    L.call('bts_dp_comm_init', ctypes.byref(comm), 1, ctypes.cast(uid, ctypes.c_void_p), 0)
    assert comm.value
    st = torch.cuda.Stream()
    sp = ctypes.c_void_p(st.cuda_stream)
    try:
        g = torch.Generator().manual_seed(3)
        flat = torch.randn(1 << 20, generator=g).to(dev)
        want = flat.clone()
        offs = (ctypes.c_long * 3)(0, 300000, 700000)
        lens = (ctypes.c_long * 3)(300000, 400000, (1 << 20) - 700000)
        st.wait_stream(torch.cuda.current_stream())
        L.call('bts_dp_allreduce_buckets', comm, ctypes.c_void_p(flat.data_ptr()), offs, lens, 3, sp)
        L.call('bts_dp_broadcast_params', comm, ctypes.c_void_p(flat.data_ptr()), flat.numel(), 0, sp)
        sums = torch.arange(13, dtype=torch.float64, device=dev) * 0.5
        L.call('bts_dp_allreduce_small', comm, ctypes.c_void_p(sums.data_ptr()), 13, sp)
        st.synchronize()
        assert torch.equal(flat, want)
        assert torch.equal(sums.cpu(), torch.arange(13, dtype=torch.float64) * 0.5)
        # argument validation: status codes, nothing enqueued
        assert L._bts_dp_allreduce_buckets(comm, ctypes.c_void_p(flat.data_ptr()), offs, lens, 0, sp) == -1
        bad = (ctypes.c_long * 1)(-4)
        assert L._bts_dp_allreduce_buckets(comm, ctypes.c_void_p(flat.data_ptr()), bad, lens, 1, sp) == -1
        assert L._bts_dp_allreduce_buckets(None, ctypes.c_void_p(flat.data_ptr()), offs, lens, 3, sp) == -2
        assert L._bts_dp_broadcast_params(comm, ctypes.c_void_p(flat.data_ptr()), flat.numel(), 1, sp) == -1      # root outside the group
        assert L._bts_dp_allreduce_small(comm, None, 13, sp) == -2
    finally:
        L.call('bts_dp_comm_destroy', comm)
