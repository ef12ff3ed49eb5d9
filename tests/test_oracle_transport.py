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


# ---- the float64 dopri5 oracle (oracle/transport.py: sample_ode_dopri5) and the product's host stepper against it -------------
def test_dopri5_oracle_analytic_and_controller_constants():
    from oracle.transport import sample_ode_dopri5
    x0 = torch.ones(2, 3, dtype=torch.float64)
    traj, st = sample_ode_dopri5(x0, lambda x, t: -x, 50, 1e-5, 1e-5, return_stats=True)       # the reference's default call shape
    ts = torch.linspace(0.0, 1.0, 50).double()
    assert traj.shape == (50, 2, 3) and torch.equal(traj[0], x0)
    assert float((traj[:, 0, 0] - torch.exp(-ts)).abs().max()) < 3e-5                            # within a few tolerances at all 50 save points
    assert st["evaluations"] == 2 + 6 * (len(st["accepted"]) + len(st["rejected"]))              # FSAL: 6 new stages per attempted step
    # documented controller: no accepted step is followed by a smaller one; growth <= ifactor = 10; steps are contiguous
    acc = st["accepted"]
    for (ta, ha), (tb, hb) in zip(acc, acc[1:]):
        assert abs(ta + ha - tb) < 1e-15 and hb <= 10.0 * ha * (1 + 1e-12)
    assert acc[0][0] == 0.0 and acc[-1][0] + acc[-1][1] >= 1.0                                  # the last step is not clipped to t = 1
    # tolerance is honoured (the end point is INTERPOLATED inside an unclipped last step, so the quartic interpolant's own
    # O(h^5) error floors the gain at tight tolerances)
    e = []
    for tol in (1e-4, 1e-7, 1e-10):
        out = sample_ode_dopri5(x0, lambda x, t: 3.0 * t.view(-1, 1).double() * x, 2, tol, tol)
        e.append(abs(float(out[-1, 0, 0]) - 4.4816890703380645))
    assert e[0] > 50 * e[1] and e[1] > 10 * e[2] and e[2] < 1e-7
    # a rejected step shrinks by at least 0.2 and re-starts from the same point
    f = lambda x, t: -50.0 * (x - torch.cos(6.0 * t.view(-1, 1).double()))
    _, st = sample_ode_dopri5(torch.tensor([[0.0], [2.0]], dtype=torch.float64), f, 3, 1e-7, 1e-7, return_stats=True)
    assert len(st["rejected"]) >= 1
    steps = sorted(st["accepted"] + st["rejected"], key=lambda p: (p[0], -p[1]))
    for tr, hr in st["rejected"]:
        nxt = [h for (t, h) in steps if t == tr and h < hr]
        assert nxt and 0.2 * hr * (1 - 1e-12) <= max(nxt) < hr


def test_host_dopri5_reproduces_the_oracle_step_sequence():
    """The product's host-driven stepper (scldm_amd.transport.Sampler, float64 state here) takes exactly the oracle's accepted and
    rejected steps and returns its trajectory, on a smooth field, one with rejections, and a nonlinear coupled one."""
    from oracle.transport import sample_ode_dopri5
    A = torch.tensor([[0.0, 2.0, -1.0], [-2.0, 0.0, 0.5], [1.0, -0.5, -0.3]], dtype=torch.float64)
    fields = [
        (lambda x, t: -x, torch.ones(3, 2, 3, dtype=torch.float64), 50, 1e-5),
        # a narrow source term at t = 0.5: the controller runs into it with a large step and has to reject (a stiffness-limited
        # field is no use here: its step sequence amplifies the last-bit differences of the two summation orders)
        (lambda x, t: -x + 10.0 * torch.exp(-((t.view(-1, 1, 1).double() - 0.5) / 0.02) ** 2), torch.tensor([[[0.0]], [[2.0]]], dtype=torch.float64), 7, 1e-7),
        (lambda x, t: torch.tanh(x @ A) * (1.0 + t.view(-1, 1, 1).double()) - 0.1 * x ** 3,
         torch.linspace(-1.5, 1.5, 24, dtype=torch.float64).view(2, 4, 3), 50, 1e-5),
    ]
    for fn_, x0, n, tol in fields:
        ref, st = sample_ode_dopri5(x0, fn_, n, tol, tol, return_stats=True)
        m = _Model(fn_)
        fn = _dopri5(n, tol, tol)
        out = fn(x0, m)
        ls = fn.last_stats
        assert ls["evaluations"] == st["evaluations"] == m.calls
        if "exp" in fn_.__code__.co_names:
            assert len(st["rejected"]) >= 2
        assert len(ls["accepted_steps"]) == len(st["accepted"]) and len(ls["rejected_steps"]) == len(st["rejected"])
        for (ta, ha), (tb, hb) in zip(ls["accepted_steps"] + ls["rejected_steps"], st["accepted"] + st["rejected"]):
            assert abs(ta - tb) <= 1e-6 * max(1.0, abs(tb)) and abs(ha - hb) <= 1e-6 * hb
        assert out.shape == ref.shape and float((out - ref).abs().max()) < 1e-9 * max(1.0, float(ref.abs().max()))


def test_dopri5_step_arithmetic_matches_scipy_rk45():
    """torchdiffeq is absent, but SciPy ships an independent implementation of the same published pair (Dormand & Prince 1980;
    `scipy.integrate.RK45`) and of the same starting-step rule (Hairer-Norsett-Wanner II.4): the oracle's stage points, stage slopes
    and 5th-order solution agree with `rk_step` to rounding, its error estimate is Shampine's 2/3 multiple of the classic embedded
    difference (torchdiffeq's `c_error`), its starting step equals `select_initial_step`, and the mid-point weights are 5th-order.
    What stays unpinned after this: the accept / grow rule and the interpolation at the save times (torchdiffeq-specific)."""
    import numpy as np
    from scipy.integrate._ivp.common import select_initial_step
    from scipy.integrate._ivp.rk import RK45, rk_step
    from oracle.transport import dopri5_initial_step, dopri5_step
    rng = np.random.default_rng(3)
    W = rng.standard_normal((12, 12)) * 0.7
    f = lambda t, y: np.tanh(W @ y) * (1.0 + 0.5 * np.sin(3.0 * t)) - 0.3 * y
    for ta, dt in ((0.0, 0.05), (0.31, 0.2), (0.9, 0.4)):
        y0 = rng.standard_normal(12)
        f0 = f(ta, y0)
        y1, err, k = dopri5_step(f, ta, dt, y0, f0)
        K = np.empty((7, 12))
        y1_s, f1_s = rk_step(f, ta, y0, f0, dt, RK45.A, RK45.B, RK45.C, K)
        assert np.abs(y1 - y1_s).max() < 1e-14 and np.abs(k - K).max() < 1e-13
        assert np.abs(k[6] - f1_s).max() < 1e-13                                   # FSAL: the 7th slope is f(t1, y1)
        assert np.abs(err - (-2.0 / 3.0) * dt * (K.T @ RK45.E)).max() < 1e-15
        for atol, rtol in ((1e-5, 1e-5), (1e-8, 1e-3)):
            h_s = select_initial_step(f, ta, y0, np.inf, np.inf, f0, 1, RK45.error_estimator_order, rtol, atol)
            assert abs(dopri5_initial_step(f, ta, y0, f0, atol, rtol) - h_s) <= 1e-15 * h_s
    # mid-point weights: y0 + dt * (c_mid . k) is the solution at ta + dt / 2 to O(dt^5) (halving dt shrinks the defect ~ 32 x)
    from oracle.transport import _DP_C_MID
    g = lambda t, y: np.array([y[1], -y[0]])                                      # (cos t, -sin t)
    defect = []
    for dt in (0.2, 0.1, 0.05):
        y0 = np.array([1.0, 0.0])
        _, _, k = dopri5_step(g, 0.0, dt, y0, g(0.0, y0))
        ym = y0 + (dt * _DP_C_MID) @ k
        defect.append(np.abs(ym - np.array([np.cos(dt / 2), -np.sin(dt / 2)])).max())
    assert defect[0] / defect[1] > 20 and defect[1] / defect[2] > 20 and defect[2] < 1e-9
