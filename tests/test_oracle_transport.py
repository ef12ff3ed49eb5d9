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
