"""The C-ABI exchange step (include/nbci.h nbci_comm_* / nbci_allreduce_bucket: RCCL behind dlopen) on the one GPU a test box has:
a world-size-1 communicator all-reduces a bucket in place (f32 and bf16), and the CU-occupancy measurement aid runs and ends."""
import ctypes as C

import pytest
import torch

pytestmark = pytest.mark.gpu


def test_cabi_allreduce_world1_and_occupy_kernel():
    from llm_bci_amd._lib import check, lib
    l = lib()
    uid = (C.c_char * 128)()
    check(l.nbci_comm_unique_id(uid), "unique_id")
    comm = C.c_void_p()
    check(l.nbci_comm_create(C.byref(comm), 1, 0, uid), "comm_create")
    s = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    x = torch.randn(1 << 20, device="cuda")
    ref = x.clone()
    check(l.nbci_allreduce_bucket(comm, C.c_void_p(x.data_ptr()), x.numel(), 0, s), "allreduce f32")
    xb = ref.bfloat16()
    check(l.nbci_allreduce_bucket(comm, C.c_void_p(xb.data_ptr()), xb.numel(), 1, s), "allreduce bf16")
    torch.cuda.synchronize()
    assert torch.equal(x, ref) and torch.equal(xb, ref.bfloat16())      # one rank: SUM = identity
    assert l.nbci_allreduce_bucket(comm, None, 4, 0, s) != 0             # bad arguments are an error code, never an abort
    l.nbci_comm_destroy(comm)
    # parked workgroups leave on their own (bounded spin): 16 CUs for 200 us beside a GEMM on the main stream
    side = torch.cuda.Stream()
    check(l.nbci_debug_occupy_cus(16, 160 * 1024, 200.0, C.c_void_p(side.cuda_stream)), "occupy")
    a = torch.randn(2048, 2048, device="cuda")
    b = a @ a
    torch.cuda.synchronize()
    assert torch.isfinite(b).all()
    check(l.nbci_set_available_cus(224), "available_cus")
    check(l.nbci_set_available_cus(256), "available_cus")
    assert l.nbci_set_available_cus(0) != 0
