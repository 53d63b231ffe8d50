"""RCCL through the C ABI (tr1_rccl_*, SURVEY 8b last row) on the GPU box: a single-rank communicator is all one GPU allows (RCCL refuses two ranks on
one device), but it runs the real librccl kernels on the real stream: init, all-reduce (bf16 wire format + fp32), reduce-scatter, all-gather."""
import os

import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.timeout(300)
def test_rccl_single_rank_collectives_through_the_c_abi(hip_ops):
    os.environ.setdefault("NCCL_SOCKET_IFNAME", "lo")
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    from time_r1_amd import rccl
    assert rccl.version() > 20000, rccl.version()
    comm = rccl.RcclComm(rccl.unique_id(), 1, 0, device=0)
    try:
        g = torch.randn(1 << 20, device="cuda:0").to(torch.bfloat16)
        ref = g.clone()
        comm.all_reduce_(g)
        f = torch.randn(4099, device="cuda:0")
        reff = f.clone()
        comm.all_reduce_(f)
        out = torch.empty(1 << 20, dtype=torch.bfloat16, device="cuda:0")
        comm.reduce_scatter(out, ref)
        gat = torch.empty(4099, device="cuda:0")
        comm.all_gather(gat, reff)
        torch.cuda.synchronize()
        assert torch.equal(g, ref) and torch.equal(f, reff) and torch.equal(out, ref) and torch.equal(gat, reff)
    finally:
        comm.close()
