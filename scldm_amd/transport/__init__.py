"""Flow-matching transport with the reference's API (src/scldm/transport/), Linear path + velocity model.

`create_transport(...)`, `Transport.training_losses`, `Sampler(transport).sample_ode(...)` keep the reference
signatures (transport/__init__.py:6-12, transport.py:110,324-332).  Differences, by design:
  * only (path_type="Linear", prediction="velocity") exists - the one combination every reference config
    uses (ldm_base.yaml:30-35); anything else raises NotImplementedError;
  * the reference hands stepping to third-party torchdiffeq (integrators.py:111, default dopri5); here the
    fixed-grid "euler" / "heun" schemes are built in (definition + KAT: oracle/transport.py), and when the
    model is a scldm_amd DiT bound through `forward_with_cfg` the whole loop runs inside one C call
    (scldm_sample_ode); the reference's default, adaptive dopri5, is a host-driven Dormand-Prince 5(4) over the
    fused `forward_with_cfg` (parity unpinned like the fixed grids: torchdiffeq is un-vendored).
"""
from __future__ import annotations

import enum

import torch


class ModelType(enum.Enum):
    NOISE = enum.auto()
    SCORE = enum.auto()
    VELOCITY = enum.auto()


class PathType(enum.Enum):
    LINEAR = enum.auto()
    GVP = enum.auto()
    VP = enum.auto()


class WeightType(enum.Enum):
    NONE = enum.auto()
    VELOCITY = enum.auto()
    LIKELIHOOD = enum.auto()


def _expand_like(t: torch.Tensor, x: torch.Tensor) -> torch.Tensor:
    return t.view(t.shape[0], *([1] * (x.dim() - 1)))


def _stream() -> int:
    return torch.cuda.current_stream().cuda_stream


class _FlowMatchLoss(torch.autograd.Function):
    """loss_b = mean((pred - ut)^2) over everything but the batch axis (mean_flat, utils.py:15-17) with its gradient w.r.t. pred,
    one HIP kernel each (scldm_fm_loss / scldm_fm_loss_bwd) instead of the ~8 elementwise / reduction launches of the eager form."""

    @staticmethod
    def forward(ctx, pred, ut):
        import ctypes as C
        from .. import _lib
        pred, ut = pred.contiguous(), ut.contiguous()
        n, e = pred.shape[0], pred[0].numel()
        loss = torch.empty(n, dtype=torch.float32, device=pred.device)
        with torch.cuda.device(pred.device):
            _lib.check(_lib.lib().scldm_fm_loss(pred.data_ptr(), ut.data_ptr(), loss.data_ptr(), n, e, _stream()), "scldm_fm_loss")
        ctx.save_for_backward(pred, ut)
        return loss

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, gloss):
        from .. import _lib
        pred, ut = ctx.saved_tensors
        n, e = pred.shape[0], pred[0].numel()
        gloss = gloss.contiguous().float()
        dpred = torch.empty_like(pred)
        with torch.cuda.device(pred.device):
            _lib.check(_lib.lib().scldm_fm_loss_bwd(pred.data_ptr(), ut.data_ptr(), gloss.data_ptr(), dpred.data_ptr(), n, e, _stream()),
                       "scldm_fm_loss_bwd")
        return dpred, None


class Transport:
    def __init__(self, *, model_type, path_type, loss_type, train_eps, sample_eps):
        if path_type is not PathType.LINEAR or model_type is not ModelType.VELOCITY:
            raise NotImplementedError("only the Linear path with a velocity model is on the hot path (ldm_base.yaml:30-35)")
        self.model_type, self.path_type, self.loss_type = model_type, path_type, loss_type
        self.train_eps, self.sample_eps = train_eps, sample_eps

    def check_interval(self, *a, **k):
        return 0, 1  # velocity + Linear integrates over exactly [0, 1] (transport.py:86-90)

    def sample(self, x1: torch.Tensor):
        """x0 ~ N(0, I), t ~ U[0, 1] (transport.py:97-108)."""
        x0 = torch.randn_like(x1)
        if x1.is_cuda and torch.cuda.is_current_stream_capturing():
            # inside a HIP-graph capture (scldm_amd.training.GraphedTrainStep) host code does not run again at replay: t comes from
            # the device generator (whose captured draws advance per replay); same distribution as the reference's CPU draw
            t = torch.rand((x1.shape[0],), device=x1.device, dtype=x1.dtype)
        elif x1.is_cuda:
            # same CPU generator draw as the reference's `torch.rand((B,)).to(x1)`, but through pinned memory and an asynchronous
            # copy: a pageable host-to-device copy blocks the host until everything queued on the stream has finished, which
            # serialises the host's kernel enqueueing with the previous step's device work
            t = torch.rand((x1.shape[0],), pin_memory=True).to(x1, non_blocking=True)
        else:
            t = torch.rand((x1.shape[0],)).to(x1)
        return t, x0, x1

    def training_losses(self, model, x1, model_kwargs=None):
        """{"pred", "loss"} with loss_b = mean((model(xt, t) - (x1 - x0))^2) (transport.py:110-150, path.py:148-151)."""
        model_kwargs = model_kwargs or {}
        t, x0, x1 = self.sample(x1)
        fused = (x1.is_cuda and x1.dtype == torch.float32 and x0.dtype == torch.float32 and t.dtype == torch.float32
                 and not x1.requires_grad and not x0.requires_grad and not t.requires_grad)
        if fused:
            # same arithmetic as below, two launches instead of five (xt bit-identical: every intermediate rounded separately)
            from .. import _lib
            x1c, x0c, tc = x1.contiguous(), x0.contiguous(), t.contiguous()
            xt, ut = torch.empty_like(x1c), torch.empty_like(x1c)
            with torch.cuda.device(x1.device):
                _lib.check(_lib.lib().scldm_fm_mix(x1c.data_ptr(), x0c.data_ptr(), tc.data_ptr(), xt.data_ptr(), ut.data_ptr(),
                                                   x1c.shape[0], x1c[0].numel(), _stream()), "scldm_fm_mix")
        else:
            te = _expand_like(t, x1)
            xt = te * x1 + (1 - te) * x0
            ut = x1 - x0
        pred = model(xt, t, **model_kwargs)
        assert pred.shape == xt.shape
        if fused and pred.is_cuda and pred.dtype == torch.float32:
            return {"pred": pred, "loss": _FlowMatchLoss.apply(pred, ut)}
        return {"pred": pred, "loss": ((pred - ut) ** 2).mean(dim=list(range(1, pred.dim())))}

    def get_drift(self):
        def body_fn(x, t, model, **model_kwargs):
            out = model(x, t, **model_kwargs)
            assert out.shape == x.shape, "Output shape from ODE solver must match input shape"
            return out

        return body_fn


def create_transport(path_type="Linear", prediction="velocity", loss_weight=None, train_eps=None, sample_eps=None):
    """Same call as scldm.transport.create_transport; eps are forced to 0 for velocity+Linear (transport/__init__.py:55-57)."""
    model_type = {"noise": ModelType.NOISE, "score": ModelType.SCORE}.get(prediction, ModelType.VELOCITY)
    loss_type = {"velocity": WeightType.VELOCITY, "likelihood": WeightType.LIKELIHOOD}.get(loss_weight, WeightType.NONE)
    ptype = {"Linear": PathType.LINEAR, "GVP": PathType.GVP, "VP": PathType.VP}[path_type]
    return Transport(model_type=model_type, path_type=ptype, loss_type=loss_type, train_eps=0, sample_eps=0)


# Dormand-Prince 5(4) tableau (Dormand & Prince 1980) and the mid-point weights of the quartic dense output used by
# torchdiffeq's dopri5 (the solver behind the reference's default `sample_ode()`, integrators.py:111).
_DP_A = ((1 / 5,), (3 / 40, 9 / 40), (44 / 45, -56 / 15, 32 / 9), (19372 / 6561, -25360 / 2187, 64448 / 6561, -212 / 729),
         (9017 / 3168, -355 / 33, 46732 / 5247, 49 / 176, -5103 / 18656), (35 / 384, 0.0, 500 / 1113, 125 / 192, -2187 / 6784, 11 / 84))
_DP_C = (1 / 5, 3 / 10, 4 / 5, 8 / 9, 1.0, 1.0)
_DP_B = (35 / 384, 0.0, 500 / 1113, 125 / 192, -2187 / 6784, 11 / 84, 0.0)
_DP_E = (35 / 384 - 1951 / 21600, 0.0, 500 / 1113 - 22642 / 50085, 125 / 192 - 451 / 720, -2187 / 6784 - -12231 / 42400,
         11 / 84 - 649 / 6300, -1.0 / 60.0)
_DP_MID = (6025192743 / 30085553152 / 2, 0.0, 51252292925 / 65400821598 / 2, -2691868925 / 45128329728 / 2,
           187940372067 / 1594534317056 / 2, -1776094331 / 19743644256 / 2, 11237099 / 235043384 / 2)


def _rms(v: torch.Tensor) -> float:
    return float(v.double().pow(2).mean().sqrt())


class Sampler:
    def __init__(self, transport: Transport):
        self.transport = transport
        self.drift = transport.get_drift()

    def _sample_dopri5(self, num_steps: int, atol: float, rtol: float):
        """Adaptive Dormand-Prince 5(4) - what `torchdiffeq.odeint(method="dopri5")` computes for the reference's default
        `sample_ode()` (integrators.py:100-112; models.py:793).  The driver follows torchdiffeq's documented scheme step for
        step (oracle/transport.py: sample_ode_dopri5 is its float64 restatement and the checker): FSAL tableau, mixed rms error
        ratio over the whole state (all cells share one step size), accept iff ratio <= 1, next step
        dt * min(10, max(0.9 / ratio^(1/5), 0.2 - or 1 after an accepted step)), automatic initial step, steps never clipped to
        the save times (the solver steps past them, past t = 1 too) and the quartic interpolant of the last accepted step
        evaluated at the float32 `linspace(0, 1, num_steps)` save times.  PARITY UNPINNED at torchdiffeq itself (un-vendored,
        unpinned).  State arithmetic in the state's dtype on its device; times and step sizes are host float64; every step
        costs one host read of the error ratio.  `fn.last_stats` = evaluations, accepted / rejected (t0, dt) lists."""
        drift = self.drift

        @torch.no_grad()
        def _sample(x, model, **model_kwargs):
            n_eval = 0

            def f(xc, tval):
                # one scalar broadcast to (B,) as a stride-0 view: any model reads it as the reference's `ones(B) * t`
                # (integrators.py:103-104), and a scldm_amd DiT can tell without a device round trip that t is uniform
                nonlocal n_eval
                n_eval += 1
                tv = torch.full((), float(tval), device=xc.device, dtype=torch.float32).expand(xc.shape[0])
                return drift(xc, tv, model, **model_kwargs)

            def comb(base, w, ks, h):       # base + h * sum_i w_i k_i, skipping structural zeros
                for wi, k in zip(w, ks):
                    if wi != 0.0:
                        base = base + (h * wi) * k
                return base

            # Device state arithmetic (round 5): on a CUDA fp32 state the stage points, the error ratio and the dense output are four
            # HIP kernels (scldm_rk_*: csrc/ode_rk.hpp) that perform the SAME fp32 operations in the same order as the torch
            # expressions of the host-composed path below (kept for CPU tensors, i.e. the oracle-side tests): ~40 elementwise
            # launches per step -> 4, the error ratio still read back once per step.
            dev = x.is_cuda and x.dtype == torch.float32 and x.numel() % 4 == 0
            if dev:
                import ctypes as C
                import numpy as np
                from .. import _lib
                L = _lib.lib()
                x = x.contiguous()
                n_el = x.numel()
                ws = torch.zeros(1024, dtype=torch.float64, device=x.device)
                f32 = lambda v: float(np.float32(v))

                def _tab(w, ks, h):
                    pairs = [(k, f32(h * wi)) for wi, k in zip(w, ks) if wi != 0.0]
                    ptrs = (C.c_void_p * len(pairs))(*[k.data_ptr() for k, _ in pairs])
                    return ptrs, (C.c_float * len(pairs))(*[c for _, c in pairs]), len(pairs), [k for k, _ in pairs]

                def _st():
                    return C.c_void_p(torch.cuda.current_stream(x.device).cuda_stream)

                def comb(base, w, ks, h):
                    ks = [k.contiguous() for k in ks]
                    ptrs, cs, nk, keep = _tab(w, ks, h)
                    out = torch.empty_like(base)
                    with torch.cuda.device(x.device):
                        _lib.check(L.scldm_rk_combine(out.data_ptr(), base.data_ptr(), ptrs, cs, nk, n_el, _st()), "scldm_rk_combine")
                    return out

                def error_ratio(y0_, y1_, ks, h):
                    ptrs, cs, nk, keep = _tab(_DP_E, ks, h)
                    with torch.cuda.device(x.device):
                        _lib.check(L.scldm_rk_error(y0_.data_ptr(), y1_.data_ptr(), ptrs, cs, nk, n_el, atol, rtol, ws.data_ptr(), _st()), "scldm_rk_error")
                    return float(ws[1022]) ** 0.5                      # the step's one host read

                def dense_fit(y0_, y1_, ks, h):
                    ptrs, cs, nk, keep = _tab(_DP_MID, ks, h)
                    cf = [torch.empty_like(y0_) for _ in range(4)]
                    with torch.cuda.device(x.device):
                        _lib.check(L.scldm_rk_dense(y0_.data_ptr(), y1_.data_ptr(), ptrs, cs, nk, ks[0].data_ptr(), ks[6].data_ptr(), f32(h), f32(2 * h),
                                                    n_el, cf[0].data_ptr(), cf[1].data_ptr(), cf[2].data_ptr(), cf[3].data_ptr(), _st()), "scldm_rk_dense")
                    return (y0_, *cf)

                def dense_eval(coeff, s_):
                    out = torch.empty_like(coeff[0])
                    with torch.cuda.device(x.device):
                        _lib.check(L.scldm_rk_poly(out.data_ptr(), *[c.data_ptr() for c in coeff], f32(s_), f32(s_ * s_), f32(s_ * s_ * s_),
                                                   f32(s_ * s_ * s_ * s_), n_el, _st()), "scldm_rk_poly")
                    return out
            else:
                def error_ratio(y0_, y1_, ks, h):
                    err = sum((h * e) * k for e, k in zip(_DP_E, ks) if e != 0.0)
                    return _rms(err / (atol + rtol * torch.maximum(y0_.abs(), y1_.abs())))

                def dense_fit(y0_, y1_, ks, h):
                    ymid = comb(y0_, _DP_MID, ks, h)
                    fa, fb = ks[0], ks[6]
                    return (y0_, h * fa, h * (fb - 4 * fa) - 11 * y0_ - 5 * y1_ + 16 * ymid,
                            h * (5 * fa - 3 * fb) + 18 * y0_ + 14 * y1_ - 32 * ymid, 2 * h * (fb - fa) - 8 * (y1_ + y0_) + 16 * ymid)

                def dense_eval(coeff, s_):
                    total, sp = coeff[0] + s_ * coeff[1], s_
                    for cf in coeff[2:]:
                        sp = sp * s_
                        total = total + sp * cf
                    return total

            ts = [float(v) for v in torch.linspace(0.0, 1.0, num_steps).double()]   # integrators.py:95, cast as the solver does
            y0 = x
            f0 = f(y0, ts[0])
            # (the first evaluation checked the packed weights against the parameters; no parameter changes inside a solve, so the
            # remaining ~110 evaluations skip the device-side fingerprint pass: scldm_amd.nnets.weights_unchanged)
            from ..nnets import weights_unchanged
            with weights_unchanged():
                return _solve(f, comb, error_ratio, dense_fit, dense_eval, ts, x, y0, f0, lambda: n_eval)

        def _solve(f, comb, error_ratio, dense_fit, dense_eval, ts, x, y0, f0, n_eval_now):
            dev = x.is_cuda and x.dtype == torch.float32 and x.numel() % 4 == 0
            # initial step (Hairer / Norsett / Wanner II.4, rms norm, exponent 1 / 5)
            scale = atol + rtol * y0.abs()
            d0, d1 = _rms(y0 / scale), _rms(f0 / scale)
            h0 = 1e-6 if (d0 < 1e-5 or d1 < 1e-5) else 0.01 * d0 / d1
            d2 = _rms((f(y0 + h0 * f0, ts[0] + h0) - f0) / scale) / h0
            h1 = max(1e-6, h0 * 1e-3) if (d1 <= 1e-15 and d2 <= 1e-15) else (0.01 / max(d1, d2)) ** (1.0 / 5.0)
            h = min(100.0 * h0, h1)
            t0 = t1 = ts[0]
            coeff = None
            out = [x]
            accepted, rejected = [], []
            for next_t in ts[1:]:
                while next_t > t1:            # advance until the save time lies inside the last accepted step
                    if not (t1 + h > t1) or n_eval_now() > 100000:
                        raise RuntimeError("dopri5: step size underflow / too many evaluations")
                    ta = t1
                    ks = [f0]
                    for a_row, c in zip(_DP_A, _DP_C):
                        yi = comb(y0, a_row, ks, h)
                        ks.append(f(yi, ta + h if c == 1.0 else ta + c * h))
                    y1 = yi                   # the last stage point is the 5th-order solution (FSAL: ks[6] = f(ta + h, y1))
                    if dev:
                        ks = [k.contiguous() for k in ks]
                    ratio = error_ratio(y0, y1, ks, h)
                    if ratio <= 1.0:
                        accepted.append((ta, h))
                        coeff = dense_fit(y0, y1, ks, h)
                        t0, t1, y0, f0 = ta, ta + h, y1, ks[6]
                    else:
                        rejected.append((ta, h))
                    if ratio == 0.0:
                        h = h * 10.0
                    else:
                        h = h * min(10.0, max(0.9 / ratio ** 0.2, 1.0 if ratio < 1.0 else 0.2))
                out.append(dense_eval(coeff, (next_t - t0) / (t1 - t0)))
            _sample.last_stats = {"evaluations": n_eval_now(), "rejected": len(rejected), "accepted_steps": accepted, "rejected_steps": rejected}
            return torch.stack(out)

        return _sample

    def sample_ode(self, *, sampling_method="dopri5", num_steps=50, atol=1e-5, rtol=1e-5, reverse=False):
        """Returns fn(x, model, **model_kwargs) -> (num_steps, *x.shape) trajectory; callers take [-1] (models.py:812).

        `num_steps` grid points = num_steps-1 steps (integrators.py:95).  The model sees t broadcast to a (B,)
        vector (integrators.py:103-104) - here as a stride-0 view of one device scalar, which a scldm_amd DiT
        recognises as uniform and shares the conditioning work across the batch.
        """
        method = sampling_method.lower()
        if method not in ("euler", "heun", "dopri5"):
            raise NotImplementedError(f"sampling_method={sampling_method!r}: 'euler', 'heun' (fixed grid) and 'dopri5' (adaptive) are provided")
        if reverse:
            raise NotImplementedError("reverse-time ODE has no caller in the reference")
        if method == "dopri5":
            return self._sample_dopri5(num_steps, atol, rtol)
        drift = self.drift

        @torch.no_grad()
        def _sample(x, model, **model_kwargs):
            ts = torch.linspace(0.0, 1.0, num_steps)
            traj = [x]

            def f(xc, tval):
                # one scalar broadcast to (B,) as a stride-0 view: any model reads it as the reference's `ones(B) * t`
                # (integrators.py:103-104), and a scldm_amd DiT can tell without a device round trip that t is uniform
                tv = torch.full((), float(tval), device=xc.device, dtype=torch.float32).expand(xc.shape[0])
                return drift(xc, tv, model, **model_kwargs)

            from ..nnets import weights_unchanged
            import contextlib
            with contextlib.ExitStack() as stack:
                for i in range(num_steps - 1):
                    h = float(ts[i + 1] - ts[i])
                    k1 = f(x, ts[i])
                    if i == 0:    # the first evaluation checked the packed weights against the parameters: the rest of the loop skips that pass
                        stack.enter_context(weights_unchanged())
                    if method == "euler":
                        x = x + h * k1
                    else:
                        k2 = f(x + h * k1, ts[i + 1])
                        x = x + (0.5 * h) * (k1 + k2)
                    traj.append(x)
            return torch.stack(traj)

        return _sample
