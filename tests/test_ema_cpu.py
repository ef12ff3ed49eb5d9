"""Host logic of scldm_amd.ema.EMA against the plain restatement of ema-pytorch 0.7.7's schedule (oracle/ema.py: parity unpinned, the
package is absent from the image) and the API / state_dict surface LatentDiffusion relies on (src/scldm/models.py:83-87,446-453,690)."""
import pytest
import torch
import torch.nn as nn

from oracle.ema import run, schedule
from scldm_amd.ema import EMA, EMA_COPY, EMA_LERP, EMA_NONE

CODES = {"none": EMA_NONE, "copy": EMA_COPY, "lerp": EMA_LERP}


@pytest.mark.parametrize("kw", [dict(beta=0.9999, update_every=10, update_after_step=10_000),       # ldm_base.yaml:51-55 (first 120 steps)
                                dict(beta=0.9999, update_every=10, update_after_step=30),
                                dict(beta=0.9, update_every=2, update_after_step=5),
                                dict(beta=0.999, update_every=1, update_after_step=0, power=0.75, inv_gamma=2.0, min_value=0.3)])
def test_schedule_matches_the_restated_package(kw):
    ema = EMA(nn.Linear(3, 2), **kw)
    ref = schedule(120, **kw)
    for step, (action, w) in enumerate(ref):
        a, ww = ema.next_action()
        assert a == CODES[action], (step, a, action)
        if action == "lerp":
            assert ww == w, (step, ww, w)
    assert ema._host_step == 120


def test_eager_update_follows_the_online_model_and_state_dict_has_the_package_keys():
    torch.manual_seed(0)
    net = nn.Linear(4, 3)
    kw = dict(beta=0.9, update_every=2, update_after_step=5)
    ema = EMA(model=net, allow_different_devices=True, use_foreach=True, **kw)
    assert all(not p.requires_grad for p in ema.ema_model.parameters())
    traj = []
    for _ in range(40):
        with torch.no_grad():
            net.weight.add_(torch.randn_like(net.weight) * 0.1)
        ema.update()
        traj.append(net.weight.detach().clone())
    ref = run(traj, **kw)
    assert torch.equal(ema.ema_model.weight, ref[-1])
    x = torch.randn(2, 4)
    assert torch.equal(ema(x), ema.ema_model(x))                      # models.py:690 calls the EMA object like the model
    sd = ema.state_dict()
    assert {"initted", "step", "ema_model.weight", "ema_model.bias", "online_model.weight", "online_model.bias"} == set(sd)
    assert int(sd["step"]) == 40 and bool(sd["initted"])
    ema2 = EMA(model=nn.Linear(4, 3), **kw)
    ema2.load_state_dict(sd)
    assert ema2._host_step == 40 and ema2._host_initted and torch.equal(ema2.ema_model.weight, ema.ema_model.weight)
    with pytest.raises(NotImplementedError):
        EMA(nn.Linear(2, 2), update_model_with_ema_every=5)
