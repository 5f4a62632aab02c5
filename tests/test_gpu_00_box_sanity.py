"""GPU box sanity (runs first): stock PyTorch/rocBLAS only, nothing of this repo.

One box of the pool returned wrong values for large GEMMs on 2026-10-04 ('A faulty box in the
pool'; profiles/r02_faulty_box_suite.log): kernels unchanged since round 1 failed their tests there and the same
tree passed on every other box.  This check makes such a box visible before the parity tests run: a large fp32
matmul, repeated, must stay close to float64, and an elementwise + row-sum kernel must be bit-for-bit repeatable."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_stock_kernels_are_sane_on_this_box():
    torch.manual_seed(0)
    a = torch.randn(31360, 512, device="cuda")
    w = torch.randn(512, 512, device="cuda") / 512 ** 0.5
    ref = a.double() @ w.double()                    # float64 on the same device (stock dgemm), ALL 31,360 rows
    scale = ref.abs().max().item()
    first = None
    for _ in range(20):
        y = a @ w                                        # (a library GEMM may legitimately differ run to run in the last
        s = (a * 1.0001).sum(dim=1)                      #  bits -- split-K atomics -- so it is held to float64, not to itself)
        torch.cuda.synchronize()
        err = (y.double() - ref).abs().max().item() / scale
        assert err < 1e-4, "THIS GPU BOX IS FAULTY: stock fp32 matmul is off by %.2g (relative) against float64" % err
        if first is None:
            first = s.clone()
        assert torch.equal(s, first), "THIS GPU BOX IS FAULTY: a stock elementwise + row-sum kernel is not repeatable run to run"
