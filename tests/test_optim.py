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


def _resume_and_late_grad(device, **kw):
    """save -> load into a fresh optimiser -> continue, and a parameter that gets its first gradient late: FusedAdam
    must track torch.optim.Adam step for step (bias correction follows each parameter's own state['step'])."""
    def make(cls, **k):
        torch.manual_seed(0)
        ps = [torch.nn.Parameter(torch.randn(300, 2, device=device) * 0.1),
              torch.nn.Parameter(torch.randn(7, device=device) * 0.1)]
        return ps, cls([{"params": ps[:1], "lr": 0.02}, {"params": ps[1:], "lr": 1e-3, "weight_decay": 0.01}], **k)

    def feed(ps, gen, late_ok):
        ps[0].grad = torch.randn(ps[0].shape, generator=gen).to(device)
        late = torch.randn(ps[1].shape, generator=gen).to(device)
        ps[1].grad = late if late_ok else None          # second parameter: no gradient in the first steps

    (pa, oa), (pb, ob) = make(torch.optim.Adam), make(FusedAdam, **kw)
    ga, gb = torch.Generator().manual_seed(5), torch.Generator().manual_seed(5)
    for k in range(6):
        feed(pa, ga, k >= 3); feed(pb, gb, k >= 3)
        oa.step(); ob.step()
    for x, y in zip(pa, pb):
        torch.testing.assert_close(x, y, rtol=2e-6, atol=1e-7)
    assert int(torch.as_tensor(ob.state[pb[0]]["step"]).item()) == 6
    assert int(torch.as_tensor(ob.state[pb[1]]["step"]).item()) == 3     # born late: its own count
    # checkpoint: a TORCH optimiser's state dict restores into FusedAdam and the other way round
    (pa2, oa2), (pb2, ob2) = make(torch.optim.Adam), make(FusedAdam, **kw)
    with torch.no_grad():
        for dst, src in zip(pa2 + pb2, pa + pb):
            dst.copy_(src)
    import copy   # load_state_dict keeps references to same-device tensors: hand every optimiser its own copy
    oa2.load_state_dict(copy.deepcopy(ob.state_dict() if not kw.get("capturable") else oa.state_dict()))
    ob2.load_state_dict(copy.deepcopy(oa.state_dict()))
    for k in range(4):
        feed(pa2, ga, True); feed(pb2, gb, True)
        oa2.step(); ob2.step()
    for x, y in zip(pa2, pb2):
        torch.testing.assert_close(x, y, rtol=2e-6, atol=2e-6)    # steps are ~2e-2: 2e-6 abs = 1e-4 of a step
    assert int(torch.as_tensor(ob2.state[pb2[0]]["step"]).item()) == 10


def test_fused_adam_resume_and_late_gradient_cpu():
    _resume_and_late_grad("cpu")


@pytest.mark.gpu
@pytest.mark.parametrize("capturable", [False, True])
def test_fused_adam_resume_and_late_gradient_gpu(capturable):
    _resume_and_late_grad("cuda:0", capturable=capturable)


@pytest.mark.gpu
def test_capturable_counter_splits_when_a_sharer_is_skipped():
    """Two parameters born in the same step share one device counter; when one of them is skipped (grad None) while the
    other steps, its count must not advance: both follow torch.optim.Adam exactly."""
    import torch
    from shacira_amd.optim import FusedAdam
    dev = "cuda:0"
    g = torch.Generator().manual_seed(3)
    init = [torch.randn(257, generator=g), torch.randn(65, generator=g)]
    grads = [[torch.randn(257, generator=g), torch.randn(65, generator=g)] for _ in range(6)]
    def run(cls, **kw):
        ps = [torch.nn.Parameter(t.clone().to(dev)) for t in init]
        opt = cls(ps, lr=1e-2, **kw)
        for k, gs in enumerate(grads):
            ps[0].grad = gs[0].to(dev)
            ps[1].grad = None if k in (2, 3) else gs[1].to(dev)     # skipped after birth, twice
            opt.step()
        return ps, opt
    pa, oa = run(torch.optim.Adam)
    pb, ob = run(FusedAdam, capturable=True)
    for x, y in zip(pa, pb):
        torch.testing.assert_close(x, y, rtol=2e-6, atol=1e-7)
    assert int(ob.state[pb[0]]["step"].item()) == 6 and int(ob.state[pb[1]]["step"].item()) == 4
