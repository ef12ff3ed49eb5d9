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
Third-party anchor available in this image: SciPy's independent `scipy.integrate.RK45` implements the same Dormand-Prince pair
and starting-step rule - `dopri5_step` / `dopri5_initial_step` below are checked against it to rounding (stage points, slopes,
5th-order solution, the error estimate as Shampine's 2/3 multiple, the starting step).  Unpinned after that: torchdiffeq's
accept / grow rule and its interpolation at the save times.
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


# ---- adaptive Dormand-Prince 5(4): the reference's DEFAULT sampler -----------------------------------------------------------
# `Sampler.sample_ode()` with no arguments (src/scldm/models.py:793) -> sampling_method="dopri5", num_steps=50, atol=rtol=1e-5
# (src/scldm/transport/transport.py:324-331) -> `torchdiffeq.odeint(_fn, x, t, method="dopri5", atol=[atol], rtol=[rtol])`
# (src/scldm/transport/integrators.py:100-112).  torchdiffeq is un-vendored and unpinned: PARITY UNPINNED at the
# stepping arithmetic.  What follows restates torchdiffeq's documented adaptive Runge-Kutta driver (README "Keyword arguments" /
# the 0.2.x `rk_common` scheme) in float64:
#   * Dormand & Prince (1980) 5(4) tableau with FSAL; the 5th-order solution is the last stage point; error estimate
#     dt * sum_i (b_i - b*_i) k_i;
#   * mixed error norm: ratio = rms( err / (atol + rtol * max(|y0|, |y1|)) ) over the WHOLE state (one step size for all cells);
#     accept iff ratio <= 1;
#   * next step dt * min(ifactor, max(safety / ratio^(1/5), dfactor)) with safety 0.9, ifactor 10, dfactor 0.2 - and dfactor
#     replaced by 1 after an accepted step (a step never shrinks when it was accepted); ratio == 0 -> dt * ifactor;
#   * initial step by Hairer-Norsett-Wanner II.4 with the rms norm and exponent 1/5 (the driver passes order - 1);
#   * steps are NOT clipped to the requested times: the solver steps past each save point (past t = 1 as well) and evaluates
#     the quartic interpolant fitted through (y0, y_mid, y1, f0, f1) of the last accepted step, x = (t - t0) / (t1 - t0);
#   * the save times are float32 `linspace(0, 1, num_steps)` values cast to float64; the drift sees float(t) broadcast to (B,).
import numpy as _np

_DP_ALPHA = _np.array([1 / 5, 3 / 10, 4 / 5, 8 / 9, 1.0, 1.0])
_DP_BETA = [
    _np.array([1 / 5]),
    _np.array([3 / 40, 9 / 40]),
    _np.array([44 / 45, -56 / 15, 32 / 9]),
    _np.array([19372 / 6561, -25360 / 2187, 64448 / 6561, -212 / 729]),
    _np.array([9017 / 3168, -355 / 33, 46732 / 5247, 49 / 176, -5103 / 18656]),
    _np.array([35 / 384, 0, 500 / 1113, 125 / 192, -2187 / 6784, 11 / 84]),
]
_DP_C_ERROR = _np.array([35 / 384 - 1951 / 21600, 0, 500 / 1113 - 22642 / 50085, 125 / 192 - 451 / 720,
                         -2187 / 6784 - -12231 / 42400, 11 / 84 - 649 / 6300, -1. / 60.])
_DP_C_MID = _np.array([6025192743 / 30085553152 / 2, 0, 51252292925 / 65400821598 / 2, -2691868925 / 45128329728 / 2,
                       187940372067 / 1594534317056 / 2, -1776094331 / 19743644256 / 2, 11237099 / 235043384 / 2])


def _rms64(a: "_np.ndarray") -> float:
    return float(_np.sqrt(_np.mean(_np.square(_np.abs(a)))))


def dopri5_step(f, ta: float, dt: float, y0: "_np.ndarray", f0: "_np.ndarray"):
    """One attempted Dormand-Prince step from (ta, y0) with f0 = f(ta, y0): returns (y1, error estimate, the 7 stage slopes).
    The stage arithmetic (alpha / beta / 5th-order weights) is the classic DP5(4) pair - tests/test_oracle_transport.py checks
    it stage by stage against SciPy's independent `scipy.integrate.RK45` (`rk_step`); the embedded error weights are
    Shampine's (torchdiffeq's `c_error`), 2/3 of the classic pair's, checked there as that multiple."""
    k = _np.empty((7, y0.size))
    k[0] = f0
    yi = y0
    for i, (alpha, beta) in enumerate(zip(_DP_ALPHA, _DP_BETA)):
        ti = ta + dt if alpha == 1.0 else ta + alpha * dt
        yi = y0 + (beta * dt) @ k[:i + 1]
        k[i + 1] = f(ti, yi)
    # c_sol == beta[-1] and c_sol[-1] == 0: the last stage point IS y1
    return yi, (dt * _DP_C_ERROR) @ k, k


def dopri5_initial_step(f, t0: float, y0: "_np.ndarray", f0: "_np.ndarray", atol: float, rtol: float) -> float:
    """Hairer-Norsett-Wanner II.4 starting step with the rms norm and exponent 1/5 (pinned to SciPy's `select_initial_step`
    in tests/test_oracle_transport.py: the same published algorithm, an independent implementation)."""
    scale = atol + _np.abs(y0) * rtol
    d0, d1 = _rms64(y0 / scale), _rms64(f0 / scale)
    h0 = 1e-6 if (d0 < 1e-5 or d1 < 1e-5) else 0.01 * d0 / d1
    f1 = f(t0 + h0, y0 + h0 * f0)
    d2 = _rms64((f1 - f0) / scale) / h0
    h1 = max(1e-6, h0 * 1e-3) if (d1 <= 1e-15 and d2 <= 1e-15) else (0.01 / max(d1, d2)) ** (1.0 / 5.0)
    return min(100 * h0, h1)


def sample_ode_dopri5(x: torch.Tensor, model_fn, num_steps: int = 50, atol: float = 1e-5, rtol: float = 1e-5,
                      safety: float = 0.9, ifactor: float = 10.0, dfactor: float = 0.2, max_num_steps: int = 100000,
                      return_stats: bool = False):
    """`Sampler.sample_ode(sampling_method="dopri5", num_steps=N, atol=, rtol=)(x, model)` restated in float64.

    `model_fn(x64, t_vec)` is the drift on a float64 state tensor and a (B,) float32 time vector (integrators.py:103-104).
    Returns the (num_steps, *x.shape) float64 trajectory (callers index [-1], models.py:812); with return_stats also
    {"accepted": [(t0, dt), ...], "rejected": [(t0, dt), ...], "evaluations": n}."""
    shape = tuple(x.shape)
    nb = shape[0]
    n_eval = 0

    def f(t: float, y: "_np.ndarray") -> "_np.ndarray":
        nonlocal n_eval
        n_eval += 1
        tv = torch.ones(nb, dtype=torch.float32) * float(t)
        out = model_fn(torch.from_numpy(y.reshape(shape).copy()), tv)
        assert tuple(out.shape) == shape, "Output shape from ODE solver must match input shape"     # transport.py:180
        return out.detach().to(torch.float64).numpy().reshape(-1)

    ts = torch.linspace(0.0, 1.0, num_steps).to(torch.float64).numpy()     # integrators.py:95, cast as the solver does
    y0 = x.detach().to(torch.float64).numpy().reshape(-1).copy()
    f0 = f(ts[0], y0)
    dt = dopri5_initial_step(f, float(ts[0]), y0, f0, atol, rtol)

    t0 = t1 = float(ts[0])
    coeff = [y0] * 5
    accepted, rejected = [], []
    out = [y0.copy()]
    for next_t in ts[1:]:
        n_steps = 0
        while next_t > t1:                                   # advance until the save point lies inside the last accepted step
            assert n_steps < max_num_steps, "max_num_steps exceeded"
            assert t1 + dt > t1, "underflow in dt"
            ta, tb = t1, t1 + dt
            y1, err, k = dopri5_step(f, ta, dt, y0, f0)
            ratio = _rms64(err / (atol + rtol * _np.maximum(_np.abs(y0), _np.abs(y1))))
            ok = ratio <= 1
            if ok:
                accepted.append((ta, dt))
                y_mid = y0 + (dt * _DP_C_MID) @ k
                fa, fb = k[0], k[6]
                coeff = [y0,
                         dt * fa,
                         dt * (fb - 4 * fa) - 11 * y0 - 5 * y1 + 16 * y_mid,
                         dt * (5 * fa - 3 * fb) + 18 * y0 + 14 * y1 - 32 * y_mid,
                         2 * dt * (fb - fa) - 8 * (y1 + y0) + 16 * y_mid]
                t0, t1, y0, f0 = ta, tb, y1, k[6]
            else:
                rejected.append((ta, dt))
            if ratio == 0:
                dt = dt * ifactor
            else:
                dt = dt * min(ifactor, max(safety / ratio ** (1.0 / 5.0), 1.0 if ratio < 1 else dfactor))
            n_steps += 1
        xs = (next_t - t0) / (t1 - t0)
        total = coeff[0] + xs * coeff[1]
        xp = xs
        for c in coeff[2:]:
            xp = xp * xs
            total = total + xp * c
        out.append(total)
    traj = torch.from_numpy(_np.stack(out).reshape((num_steps,) + shape))
    if return_stats:
        return traj, {"accepted": accepted, "rejected": rejected, "evaluations": n_eval}
    return traj
