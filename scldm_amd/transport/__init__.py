"""Flow-matching transport with the reference's API (src/scldm/transport/), Linear path + velocity model.

`create_transport(...)`, `Transport.training_losses`, `Sampler(transport).sample_ode(...)` keep the reference
signatures (transport/__init__.py:6-12, transport.py:110,324-332).  Differences, by design:
  * only (path_type="Linear", prediction="velocity") exists - the one combination every reference config
    uses (ldm_base.yaml:30-35); anything else raises NotImplementedError;
  * the reference hands stepping to third-party torchdiffeq (integrators.py:111, default dopri5); here the
    fixed-grid "euler" / "heun" schemes are built in (definition + KAT: oracle/transport.py), and when the
    model is a scldm_amd DiT bound through `forward_with_cfg` the whole loop runs inside one C call
    (scldm_sample_ode).  Adaptive dopri5 is not provided.
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
        t = torch.rand((x1.shape[0],)).to(x1)
        return t, x0, x1

    def training_losses(self, model, x1, model_kwargs=None):
        """{"pred", "loss"} with loss_b = mean((model(xt, t) - (x1 - x0))^2) (transport.py:110-150, path.py:148-151)."""
        model_kwargs = model_kwargs or {}
        t, x0, x1 = self.sample(x1)
        te = _expand_like(t, x1)
        xt = te * x1 + (1 - te) * x0
        ut = x1 - x0
        pred = model(xt, t, **model_kwargs)
        assert pred.shape == xt.shape
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


class Sampler:
    def __init__(self, transport: Transport):
        self.transport = transport
        self.drift = transport.get_drift()

    def sample_ode(self, *, sampling_method="euler", num_steps=50, atol=1e-5, rtol=1e-5, reverse=False):
        """Returns fn(x, model, **model_kwargs) -> (num_steps, *x.shape) trajectory; callers take [-1] (models.py:812).

        `num_steps` grid points = num_steps-1 steps (integrators.py:95).  The model sees t broadcast to a (B,)
        vector (integrators.py:103-104); that vector carries a `_scldm_uniform_t` hint so a scldm_amd DiT can
        share the conditioning work across the batch.
        """
        method = sampling_method.lower()
        if method not in ("euler", "heun"):
            raise NotImplementedError(f"sampling_method={sampling_method!r}: fixed-grid 'euler' and 'heun' are built in; the reference's "
                                      "adaptive default 'dopri5' lives in third-party torchdiffeq and is not provided")
        if reverse:
            raise NotImplementedError("reverse-time ODE has no caller in the reference")
        drift = self.drift

        @torch.no_grad()
        def _sample(x, model, **model_kwargs):
            ts = torch.linspace(0.0, 1.0, num_steps)
            traj = [x]

            def f(xc, tval):
                tv = torch.full((xc.shape[0],), float(tval), device=xc.device, dtype=torch.float32)
                tv._scldm_uniform_t = True
                return drift(xc, tv, model, **model_kwargs)

            for i in range(num_steps - 1):
                h = float(ts[i + 1] - ts[i])
                k1 = f(x, ts[i])
                if method == "euler":
                    x = x + h * k1
                else:
                    k2 = f(x + h * k1, ts[i + 1])
                    x = x + (0.5 * h) * (k1 + k2)
                traj.append(x)
            return torch.stack(traj)

        return _sample
