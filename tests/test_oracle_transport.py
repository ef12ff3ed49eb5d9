"""Oracle transport: training_losses vs the reference fixture; fixed-step Euler/Heun pinned by an
analytic known-answer test (the stepping itself lives in un-vendored torchdiffeq: parity unpinned)."""
import torch

from conftest import load_golden, max_abs_rel
from oracle.transport import sample_ode_fixed, time_grid, training_losses
from test_oracle_dit import setup
from oracle.dit import dit_forward


def test_training_losses_matches_reference():
    g = load_golden("transport_tiny")
    _, cfg, sd = setup("dit_tiny")
    lab = torch.from_numpy(g["label_a"])
    out = training_losses(lambda xt, t: dit_forward(sd, cfg, xt, t, {"a": lab}), torch.from_numpy(g["x1"]),
                          torch.from_numpy(g["x0"]), torch.from_numpy(g["t"]))
    assert max_abs_rel(out["pred"], g["pred"]) < 2e-5
    assert max_abs_rel(out["loss"], g["loss"]) < 2e-5


def test_euler_kat_and_grid():
    # f = -x, n = 4 evaluations: x_end = (1 - 1/4)^4 = 0.31640625 exactly in fp32
    calls = []
    def f(x, t):
        calls.append(float(t[0]))
        return -x
    x = sample_ode_fixed(torch.ones(2, 3), f, num_steps=5, method="euler")
    assert calls == [0.0, 0.25, 0.5, 0.75]  # N grid points -> N-1 evaluations
    assert torch.equal(x, torch.full((2, 3), 0.31640625))
    assert torch.equal(time_grid(5), torch.tensor([0.0, 0.25, 0.5, 0.75, 1.0]))


def test_heun_kat():
    # f = -x: one Heun step multiplies by (1 - h + h^2/2); n=4 -> (0.78125)^4
    x = sample_ode_fixed(torch.ones(1, 1, dtype=torch.float64), lambda x, t: -x, num_steps=5, method="heun")
    assert abs(float(x) - 0.78125 ** 4) < 1e-15
    # second-order convergence on dx/dt = t*x  (exact: exp(1/2))
    errs = []
    for n in (11, 21, 41):
        x = sample_ode_fixed(torch.ones(1, 1, dtype=torch.float64), lambda x, t: t.double().view(-1, 1) * x, n, "heun")
        errs.append(abs(float(x) - 1.6487212707001282))
    assert 3.5 < errs[0] / errs[1] < 4.5 and 3.5 < errs[1] / errs[2] < 4.5


# ---- adaptive dopri5 of scldm_amd.transport.Sampler (host-side stepping logic; runs on any torch device) -------------------
class _Model:
    def __init__(self, fn):
        self.fn, self.calls = fn, 0

    def __call__(self, x, t, **kw):
        self.calls += 1
        assert t.shape == (x.shape[0],) and bool((t == t[0]).all())      # scalar t broadcast to the batch (integrators.py:103-104)
        return self.fn(x, t)


def _dopri5(num_steps, atol, rtol):
    from scldm_amd.transport import Sampler, create_transport
    return Sampler(create_transport("Linear", "velocity", "velocity", 1e-5, 1e-5)).sample_ode(
        sampling_method="dopri5", num_steps=num_steps, atol=atol, rtol=rtol)


def test_dopri5_default_method_and_analytic_solutions():
    import inspect
    from scldm_amd.transport import Sampler
    assert inspect.signature(Sampler.sample_ode).parameters["sampling_method"].default == "dopri5"   # transport.py:326
    x0 = torch.ones(3, 2, 2, dtype=torch.float64)
    m = _Model(lambda x, t: -x)
    fn = _dopri5(5, 1e-9, 1e-9)
    out = fn(x0, m)
    assert out.shape == (5, 3, 2, 2) and torch.equal(out[0], x0)
    exact = torch.exp(-torch.linspace(0, 1, 5, dtype=torch.float64))
    assert float((out[:, 0, 0, 0] - exact).abs().max()) < 1e-8          # dense output at the requested times
    assert m.calls == fn.last_stats["evaluations"] and m.calls < 200
    # tolerance is honoured: tighter tolerance -> smaller error and more evaluations
    errs, evals = [], []
    for tol in (1e-3, 1e-6, 1e-9):
        m = _Model(lambda x, t: t.view(-1, 1, 1).double() * x * 3.0)
        fn = _dopri5(2, tol, tol)
        errs.append(abs(float(fn(x0, m)[-1, 0, 0, 0]) - float(torch.exp(torch.tensor(1.5, dtype=torch.float64)))))
        evals.append(m.calls)
    assert errs[0] > errs[1] > errs[2] and errs[2] < 1e-7 and evals[0] < evals[1] < evals[2]


def test_dopri5_rejects_steps_on_a_stiff_transient_and_matches_fine_heun():
    # dx/dt = -50 (x - cos(6 t)): forces step rejections; compare with a 20 000-step Heun solve
    f = lambda x, t: -50.0 * (x - torch.cos(6.0 * t.view(-1, 1).double()))
    x0 = torch.tensor([[0.0], [2.0]], dtype=torch.float64)
    m = _Model(f)
    fn = _dopri5(3, 1e-7, 1e-7)
    out = fn(x0, m)
    ref = sample_ode_fixed(x0, lambda x, t: f(x, t), 20001, "heun")
    assert float((out[-1] - ref).abs().max()) < 1e-5
    assert fn.last_stats["rejected"] >= 1
