"""CPU restatement of the flow-matching transport pieces on the path.  TEST INFRASTRUCTURE ONLY.

Linear path + velocity prediction only (the only combination any reference config
uses: experiments/configs/model/ldm_base.yaml:30-35).  create_transport forces
train_eps = sample_eps = 0 for that combination (src/scldm/transport/__init__.py:55-57),
so the integration interval is exactly [0, 1].

PARITY UNPINNED at the stepping arithmetic: the reference hands the loop to the
un-vendored, unpinned third-party `torchdiffeq.odeint`
(src/scldm/transport/integrators.py:4,111).  We restate the published fixed-grid
schemes and pin them with an analytic known-answer test (tests/test_oracle_transport.py):
  euler:  x_{i+1} = x_i + h f(t_i, x_i)
  heun :  k1 = f(t_i, x_i); k2 = f(t_i + h, x_i + h k1); x_{i+1} = x_i + h/2 (k1 + k2)
(the same form as the reference's own SDE Heun step without noise, integrators.py:39-48).
The grid is th.linspace(0, 1, num_steps) (integrators.py:95): num_steps points ->
num_steps-1 steps; the model sees t broadcast to a (B,) vector (:103-104).
"""
from __future__ import annotations

import torch


def time_grid(num_steps: int) -> torch.Tensor:
    """ode.__init__, src/scldm/transport/integrators.py:95 with t0=0, t1=1."""
    return torch.linspace(0.0, 1.0, num_steps)


def sample_ode_fixed(x: torch.Tensor, model_fn, num_steps: int, method: str = "euler",
                     return_all: bool = False):
    """Sampler.sample_ode(sampling_method=euler|heun, num_steps=N)(x, model) restated
    (src/scldm/transport/transport.py:324-369; integrators.py:100-112).

    `model_fn(x, t_vec)` is the drift (velocity model => identity wrapper,
    transport.py:167-169).  Returns the final state (what callers index with [-1],
    src/scldm/models.py:812) or the whole (num_steps, ...) trajectory.
    """
    ts = time_grid(num_steps).to(torch.float32)
    traj = [x]
    for i in range(num_steps - 1):
        t0, t1 = ts[i], ts[i + 1]
        h = (t1 - t0).to(x.dtype)
        tv = torch.ones(x.shape[0], dtype=torch.float32) * t0
        k1 = model_fn(x, tv)
        assert k1.shape == x.shape, "Output shape from ODE solver must match input shape"  # transport.py:180
        if method == "euler":
            x = x + h * k1
        elif method == "heun":
            tv1 = torch.ones(x.shape[0], dtype=torch.float32) * t1
            k2 = model_fn(x + h * k1, tv1)
            x = x + (0.5 * h) * (k1 + k2)
        else:
            raise NotImplementedError(method)
        traj.append(x)
    return torch.stack(traj) if return_all else x


def plan_linear(t: torch.Tensor, x0: torch.Tensor, x1: torch.Tensor):
    """ICPlan.plan, src/scldm/transport/path.py:131-151: xt = t x1 + (1-t) x0, ut = x1 - x0."""
    te = t.view(-1, *([1] * (x1.dim() - 1))).to(x1.dtype)
    xt = te * x1 + (1 - te) * x0
    ut = x1 - x0
    return t, xt, ut


def training_losses(model_fn, x1: torch.Tensor, x0: torch.Tensor, t: torch.Tensor) -> dict:
    """Transport.training_losses for (Linear, velocity), src/scldm/transport/transport.py:110-150,
    with the random draws (x0 ~ N, t ~ U[0,1], transport.py:97-108) injected.
    loss = mean over non-batch dims of (model(xt, t) - ut)^2 (utils.py:15-17)."""
    _, xt, ut = plan_linear(t, x0, x1)
    pred = model_fn(xt, t)
    loss = ((pred - ut) ** 2).mean(dim=list(range(1, pred.dim())))
    return {"pred": pred, "loss": loss, "xt": xt, "ut": ut}
