"""FusedAdam (row f1) against torch.optim.Adam -- the optimiser the reference's trainers instantiate."""
import pytest
import torch

from shacira_amd.optim import FusedAdam


def _run(opt_cls, device, steps, **kw):
    torch.manual_seed(0)
    ps = [torch.nn.Parameter(torch.randn(n, 2, device=device) * 0.1) for n in (1000, 37, 4096 * 3 + 1)]
    ps.append(torch.nn.Parameter(torch.randn(5, device=device)))
    groups = [{"params": ps[:2], "lr": 0.02, "weight_decay": 0.0}, {"params": ps[2:], "lr": 1e-3, "weight_decay": 0.01}]
    opt = opt_cls(groups, eps=1e-8, **kw)
    g = torch.Generator(device="cpu").manual_seed(1)
    for _ in range(steps):
        for p in ps:
            p.grad = torch.randn(p.shape, generator=g).to(device)
        opt.step()
    return ps, opt


def test_fused_adam_cpu_formula_matches_torch():
    a, _ = _run(torch.optim.Adam, "cpu", 6)
    b, ob = _run(FusedAdam, "cpu", 6)
    for x, y in zip(a, b):
        torch.testing.assert_close(x, y, rtol=2e-6, atol=1e-7)   # updates are ~1e-2: 1e-7 abs = 1e-5 of a step
    st = next(iter(ob.state.values()))
    assert set(st) == {"step", "exp_avg", "exp_avg_sq"} and float(st["step"]) == 6


@pytest.mark.gpu
def test_fused_adam_kernel_matches_torch():
    a, _ = _run(torch.optim.Adam, "cuda:0", 8)
    b, _ = _run(FusedAdam, "cuda:0", 8)
    for x, y in zip(a, b):
        torch.testing.assert_close(x, y, rtol=2e-6, atol=1e-7)   # updates are ~1e-2: 1e-7 abs = 1e-5 of a step
    c, _ = _run(FusedAdam, "cuda:0", 3, zero_grad_in_step=True)
    assert all(float(p.grad.abs().sum()) == 0.0 for p in c)
