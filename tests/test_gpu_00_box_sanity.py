"""GPU box sanity (runs first): stock PyTorch/rocBLAS only, nothing of this repo.

One box of the pool returned wrong values for large GEMMs on 2026-10-04 ('A faulty box in the
pool'; profiles/r02_faulty_box_suite.log): kernels unchanged since round 1 failed their tests there and the same
tree passed on every other box.  This check makes such a box visible before the parity tests run: a large fp32
matmul and a large elementwise reduction, repeated, must be bit-for-bit repeatable and close to float64."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_stock_kernels_are_repeatable_on_this_box():
    torch.manual_seed(0)
    a = torch.randn(31360, 512, device="cuda")
    w = torch.randn(512, 512, device="cuda") / 512 ** 0.5
    ref = (a[:2048].double() @ w.double()).float()
    first = None
    for _ in range(20):
        y = a @ w
        s = (a * 1.0001).sum(dim=1)
        torch.cuda.synchronize()
        if first is None:
            first = (y.clone(), s.clone())
        assert torch.equal(y, first[0]) and torch.equal(s, first[1]), \
            "THIS GPU BOX IS FAULTY: a stock rocBLAS matmul / reduction is not repeatable run to run"
    err = (first[0][:2048] - ref).abs().max().item()
    assert err < 1e-3, "THIS GPU BOX IS FAULTY: stock rocBLAS fp32 matmul is off by %g against float64" % err
