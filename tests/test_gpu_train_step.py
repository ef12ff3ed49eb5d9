"""The whole flow-matching optimisation step as one C call (scldm_dit_train_step, include/scldm_hip.h):
batch preparation and loss kernels against torch compositions on the SAME draws (bit for bit), the statistics of the device-side
draws (the reference's are torch.rand / randn / randint: RNG parity is impossible by construction), the AdamW + EMA launch against
torch's lerp_ (bit for bit) and the oracle's restatement of ema-pytorch's schedule, and FusedTrainStep (eager and HIP-graph replay)
against the composed autograd route on the same draws."""
import copy
import ctypes as C

import numpy as np
import pytest
import torch

from scldm_amd import _lib
from test_gpu_train import build

pytestmark = pytest.mark.gpu


def _prepare(x1, labels, nulls, strategy, p_drop, rng, drop=1, want_x0=True):
    L = _lib.lib()
    n, e = x1.shape[0], x1[0].numel()
    f = dict(dtype=torch.float32, device="cuda")
    t, xt, ut = torch.empty(n, **f), torch.empty(n, e, **f), torch.empty(n, e, **f)
    x0 = torch.empty(n, e, **f) if want_x0 else None
    out = torch.empty(len(labels), n, dtype=torch.long, device="cuda")
    ptrs = _lib.ptr_array([None if l is None else l.data_ptr() for l in labels])
    nl = (C.c_int * len(labels))(*nulls)
    _lib.check(L.scldm_fm_prepare(x1.data_ptr(), C.cast(ptrs, _lib.c_void_pp), nl, len(labels), strategy, drop, p_drop, rng.data_ptr(), n, e,
                                  t.data_ptr(), None if x0 is None else x0.data_ptr(), xt.data_ptr(), ut.data_ptr(), out.data_ptr(),
                                  torch.cuda.current_stream().cuda_stream), "scldm_fm_prepare")
    return t, x0, xt, ut, out


def test_prepare_kernel_is_the_eager_composition_on_its_own_draws():
    """transport.py:97-108 + path.py:148-151 + nnets.py:395-402,440-452 in one kernel: xt / ut are the eager expressions of the kernel's
    own (t, x0) bit for bit; the draws have the right distributions; labels follow one mask per row; the class of a mutually_exclusive
    model is drawn per call on the device; everything is a function of (seed, step counter) only."""
    n = 4096
    g = torch.Generator(device="cuda").manual_seed(1)
    x1 = torch.randn(n, 16, 16, device="cuda", generator=g)
    a = torch.randint(0, 4, (n,), device="cuda", generator=g)
    b = torch.randint(0, 2024, (n,), device="cuda", generator=g)
    rng = torch.tensor([12345, 7], dtype=torch.int64, device="cuda")
    t, x0, xt, ut, lab = _prepare(x1, [a, b], [4, 2024], 1, 0.8, rng)
    x1f = x1.view(n, -1)
    te = t.view(-1, 1)
    assert torch.equal(xt, te * x1f + (1 - te) * x0) and torch.equal(ut, x1f - x0)
    assert float(t.min()) >= 0.0 and float(t.max()) < 1.0
    assert abs(float(t.mean()) - 0.5) < 0.02 and abs(float(t.var()) - 1 / 12) < 0.005
    z = x0.double()
    assert abs(float(z.mean())) < 0.005 and abs(float(z.var()) - 1) < 0.01 and abs(float((z ** 4).mean()) - 3) < 0.05 and abs(float((z ** 3).mean())) < 0.02
    assert abs(float(torch.corrcoef(torch.stack([z[:, 0], z[:, 1]]))[0, 1])) < 0.06        # neighbours of one Philox call are independent
    # joint: ONE mask per row over every class (nnets.py:440-443)
    da, db = lab[0] == 4, lab[1] == 2024
    assert torch.equal(da, db) and abs(float(da.float().mean()) - 0.8) < 0.03
    assert torch.equal(lab[0][~da], a[~da]) and torch.equal(lab[1][~db], b[~db])
    # same state -> same batch; next step -> another one; x0 = NULL changes nothing else
    t2, _, xt2, ut2, lab2 = _prepare(x1, [a, b], [4, 2024], 1, 0.8, rng, want_x0=False)
    assert torch.equal(t, t2) and torch.equal(xt, xt2) and torch.equal(ut, ut2) and torch.equal(lab, lab2)
    rng[1] += 1
    t3, x03, _, _, lab3 = _prepare(x1, [a, b], [4, 2024], 1, 0.8, rng)
    assert not torch.equal(t, t3) and not torch.equal(x0, x03) and not torch.equal(lab, lab3)
    # drop = 0 (eval-mode forward_with_cfg never drops): labels pass through
    assert torch.equal(_prepare(x1, [a, b], [4, 2024], 1, 0.8, rng, drop=0)[4], torch.stack([a, b]))
    # mutually_exclusive with two classes: one class per CALL (nnets.py:395), the other all null; both get their turn
    picks = []
    for step in range(64):
        rng[1] = step
        out = _prepare(x1[:256], [a[:256], b[:256]], [4, 2024], 0, 0.5, rng)[4]
        live = [bool((out[0] != 4).any()), bool((out[1] != 2024).any())]
        assert sum(live) == 1
        picks.append(live.index(True))
        c = picks[-1]
        keep = out[c] != [4, 2024][c]
        assert abs(float(keep.float().mean()) - 0.5) < 0.15 and torch.equal(out[c][keep], [a, b][c][:256][keep])
    assert 16 <= sum(picks) <= 48
    # a class missing from `condition`: never chosen, always null
    out = _prepare(x1[:64], [None, b[:64]], [4, 2024], 0, 0.0, rng)[4]
    assert bool((out[0] == 4).all()) and torch.equal(out[1], b[:64])
    with pytest.raises(_lib.ScldmError):
        _prepare(x1[:64], [None, b[:64]], [4, 2024], 1, 0.0, rng)          # joint needs every class (nnets.py:449)


def test_loss_grad_kernel_matches_the_composed_kernels_bit_for_bit():
    L = _lib.lib()
    n, e = 1000, 256
    g = torch.Generator(device="cuda").manual_seed(2)
    pred, ut = torch.randn(n, e, device="cuda", generator=g), torch.randn(n, e, device="cuda", generator=g)
    rows, mean, dpred = torch.empty(n, device="cuda"), torch.zeros((), device="cuda"), torch.empty(n, e, device="cuda")
    ticket = torch.zeros(1, dtype=torch.int32, device="cuda")
    rng = torch.tensor([5, 41], dtype=torch.int64, device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    for rep in range(3):
        _lib.check(L.scldm_fm_loss_grad(pred.data_ptr(), ut.data_ptr(), n, e, rows.data_ptr(), mean.data_ptr(), dpred.data_ptr(), ticket.data_ptr(),
                                        rng.data_ptr(), st), "scldm_fm_loss_grad")
        rows_ref, d_ref = torch.empty(n, device="cuda"), torch.empty(n, e, device="cuda")
        gl = torch.full((n,), 1.0 / n, device="cuda")
        _lib.check(L.scldm_fm_loss(pred.data_ptr(), ut.data_ptr(), rows_ref.data_ptr(), n, e, st), "scldm_fm_loss")
        _lib.check(L.scldm_fm_loss_bwd(pred.data_ptr(), ut.data_ptr(), gl.data_ptr(), d_ref.data_ptr(), n, e, st), "scldm_fm_loss_bwd")
        assert torch.equal(rows, rows_ref) and torch.equal(dpred, d_ref)
        assert abs(float(mean) - float(rows.double().mean())) <= 2e-6 * float(mean)
        assert int(ticket) == 0 and int(rng[1]) == 42 + rep                      # the ticket resets itself, the Philox step counter advances
    m0 = float(mean)
    _lib.check(L.scldm_fm_loss_grad(pred.data_ptr(), ut.data_ptr(), n, e, rows.data_ptr(), mean.data_ptr(), dpred.data_ptr(), ticket.data_ptr(), None, st), "x")
    assert float(mean) == m0 and int(rng[1]) == 44                                # fixed-order sums: repeatable; NULL state: no advance


def _torch_ema_reference(traj, kw):
    from oracle.ema import run
    return run(traj, **kw)


def test_adamw_launch_with_ema_is_torch_lerp_bit_for_bit_and_follows_lr():
    """scldm_adamw_table_step: (a) the parameters follow torch's fused AdamW under a per-step learning-rate schedule read from device
    memory; (b) the EMA tensors equal `ema.lerp_(p, 1 - decay)` / copies applied by torch on the SAME parameter trajectory, bit for bit,
    through the copy phase, lerp weights above and below 0.5 (both branches of torch.lerp) and the steps without an update;
    (c) a found_inf step skips AdamW but not the EMA hook (models.py:83-87 runs after every batch)."""
    from scldm_amd.ema import EMA
    from scldm_amd.optim import AdamW
    torch.manual_seed(0)
    shapes = [(300, 17), (4096,), (5,), (64, 64), (1, 9000)]
    net = torch.nn.ParameterList([torch.nn.Parameter(torch.randn(s, device="cuda")) for s in shapes])
    ref = torch.nn.ParameterList([torch.nn.Parameter(p.detach().clone()) for p in net])
    kw = dict(beta=0.95, update_every=2, update_after_step=7)
    ema = EMA(model=net, **kw)
    opt = AdamW(net.parameters(), lr=1e-2, weight_decay=0.05)
    opt.attach_ema(ema)
    topt = torch.optim.AdamW(ref.parameters(), lr=1e-2, weight_decay=0.05, fused=True)
    traj = [[] for _ in shapes]
    flag = torch.zeros((), device="cuda")
    for step in range(60):
        lr = 1e-2 * (0.5 + 0.5 * np.cos(step / 10))                               # LambdaLR-style schedule: a new value every step
        for g_ in opt.param_groups + topt.param_groups:
            g_["lr"] = lr
        grads = [torch.randn_like(p) for p in net]
        for p, q, g_ in zip(net, ref, grads):
            p.grad, q.grad = g_.clone(), g_.clone()
        skip = step == 37
        flag.fill_(1.0 if skip else 0.0)
        opt.found_inf, opt.grad_scale = flag, None
        opt.step()
        del opt.found_inf, opt.grad_scale
        ema.update()
        if not skip:
            topt.step()
        for i, p in enumerate(net):
            traj[i].append(p.detach().clone())
    for p, q in zip(net, ref):
        assert float((p - q).detach().abs().max()) <= 2e-6 * float(q.detach().abs().max())         # (torch's kernel forms lr-dependent terms in another order)
    for i, avg in enumerate(ema.ema_model.parameters()):
        want = _torch_ema_reference(traj[i], kw)[-1]
        assert torch.equal(avg, want), i
    sched = __import__("oracle.ema", fromlist=["schedule"]).schedule(60, **kw)
    ws = [w for a, w in sched if a == "lerp"]
    assert min(ws) < 0.5 < max(ws)                                                 # both branches of torch.lerp were exercised
    with pytest.raises(RuntimeError, match="has not stepped"):
        ema.update()                                                               # one update() per optimizer step


@pytest.mark.parametrize("strategy,vocab,prec,n", [("joint", {"cell_line": 4, "gene": 2024}, "bf16", 96), ("mutually_exclusive", {"a": 5, "b": 9}, "bf16", 96),
                                                   ("joint", {"cell_line": 4, "gene": 2024}, "fp16", 96),
                                                   ("joint", {"cell_line": 4, "gene": 2024}, "bf16", 37)])     # ragged: the last 64-token tile is padded
def test_fused_train_step_equals_the_composed_route_and_replays_as_a_graph(strategy, vocab, prec, n):
    """FusedTrainStep (one C call per step) against the autograd route (DiT.forward -> _FlowMatchLoss -> backward -> AdamW.step ->
    EMA) fed the SAME draws: gradients, parameters, optimizer state and EMA bit for bit; then the HIP-graph form against the eager
    form over several steps (device-side step counter, learning-rate schedule and EMA actions reach the replays)."""
    from scldm_amd.ema import EMA
    from scldm_amd.optim import AdamW
    from scldm_amd.training import FusedTrainStep
    from scldm_amd.transport import _FlowMatchLoss, create_transport
    m1, _, _ = build(vocab, strategy, 8, 31)
    m1.precision = prec
    m1.cfg_dropout_prob = 0.6
    m2 = copy.deepcopy(m1)
    m3 = copy.deepcopy(m1)
    tr = create_transport("Linear", "velocity", "velocity", 1e-5, 1e-5)
    kw = dict(beta=0.9, update_every=2, update_after_step=3)
    gen = torch.Generator(device="cuda").manual_seed(4)
    batches = [(torch.randn(n, 16, 16, device="cuda", generator=gen), {k: torch.randint(0, v, (n,), device="cuda", generator=gen) for k, v in vocab.items()})
               for _ in range(6)]
    lrs = [1e-3 * (1 + s) / 6 for s in range(6)]

    def make(m, graph):
        opt = AdamW([p for p in m.parameters() if p.requires_grad], lr=1e-3, weight_decay=0.01)
        ema = EMA(model=m, **kw)
        return FusedTrainStep(m, tr, opt, n, list(vocab), ema=ema, seed=99, graph=graph), opt, ema

    f1, o1, e1 = make(m1, False)
    # ---- step 0: fused eager vs composed on the same draws
    o1.param_groups[0]["lr"] = lrs[0]
    x1, cond = batches[0]
    loss1 = f1(x1, cond).clone()
    e1.update()
    t, xt, ut, lab = f1.t.clone(), f1.xt.clone(), f1.ut.clone(), f1.labels.clone()
    g1 = f1.flat.clone()
    o2 = AdamW([p for p in m2.parameters() if p.requires_grad], lr=lrs[0], weight_decay=0.01)
    e2 = EMA(model=m2, **kw)
    m2.cfg_dropout_prob = 0.0                                    # the labels below are already dropped: no second mask
    names = m2._class_names
    if strategy == "joint":
        cond2 = {c: lab[i] for i, c in enumerate(names)}
    else:                                                        # the class the kernel drew is the only one that is not all-null
        live = [i for i, c in enumerate(names) if bool((lab[i] != vocab[c]).any())]
        cond2 = {names[live[0] if live else 0]: lab[live[0] if live else 0]}
    pred = m2(xt.view(n, 16, 16), t, cond2, force_drop_ids=False)
    loss2 = _FlowMatchLoss.apply(pred.view(n, -1), ut).mean()
    loss2.backward()
    assert abs(float(loss1) - float(loss2)) <= 2e-6 * abs(float(loss2))
    offs = m1._grad_offsets
    for (k, p1), (_, p2) in zip(m1.named_parameters(), m2.named_parameters()):
        if p2.grad is None:
            continue
        assert torch.equal(g1[offs[id(p1)]:offs[id(p1)] + p1.numel()].view(p1.shape), p2.grad), k
    if prec == "fp16":
        o2.found_inf, o2.grad_scale = m2.found_inf_flag(), None
    o2.step()
    e2.update()
    for (k, p1), (_, p2) in zip(m1.named_parameters(), m2.named_parameters()):
        assert torch.equal(p1, p2), k
    for (k, a), (_, b) in zip(e1.ema_model.named_parameters(), e2.ema_model.named_parameters()):
        assert torch.equal(a, b), k
    # ---- steps 1-5 eager on m1; steps 0-5 as graph replays on m3 (same seed, same batches, same schedule): identical models
    for s in range(1, 6):
        o1.param_groups[0]["lr"] = lrs[s]
        f1(*batches[s])
        e1.update()
    f3, o3, e3 = make(m3, True)
    assert f3.graph is not None
    losses = []
    for s in range(6):
        o3.param_groups[0]["lr"] = lrs[s]
        losses.append(float(f3(*batches[s])))
        e3.update()
    assert abs(losses[0] - float(loss1)) <= 2e-6 * abs(float(loss1)) and len(set(losses)) == 6
    for (k, p1), (_, p3) in zip(m1.named_parameters(), m3.named_parameters()):
        assert torch.equal(p1, p3), k
    for (k, a), (_, b) in zip(e1.ema_model.named_parameters(), e3.ema_model.named_parameters()):
        assert torch.equal(a, b), k
    assert int(f3.rng[1]) == 6 and float(o3.param_groups[0]["_step_t"]) == 6.0
    # the EMA model is a working DiT (models.py:690 evaluates it): its packed copies follow the in-place updates
    e3.ema_model.eval()
    y = e3(batches[0][0], torch.full((n,), 0.5, device="cuda"), {k: v for k, v in batches[0][1].items()} if strategy == "joint" else
           {list(vocab)[0]: batches[0][1][list(vocab)[0]]})
    assert torch.isfinite(y).all()


def test_fused_train_step_with_the_frozen_vae_encode_in_the_graph():
    """models.py:641: every training step encodes the tokenised batch with the frozen VAE; here the encode sits inside the captured
    step (static token buffers), and the latents it trains on are the ones TransformerVAE.encode returns."""
    from scldm_amd.optim import AdamW
    from scldm_amd.training import FusedTrainStep
    from scldm_amd.transport import create_transport
    from test_gpu_vae import _fresh_vae
    n, S, G = 64, 200, 500
    vae, _, _ = _fresh_vae(G, 5)
    for p in vae.parameters():
        p.requires_grad_(False)
    vae.precision = "fp16"
    m, _, _ = build({"clusters": 14}, "mutually_exclusive", 8, 32)
    m.precision = "bf16"
    m.cfg_dropout_prob = 0.8
    opt = AdamW([p for p in m.parameters() if p.requires_grad], lr=1e-3)
    tr = create_transport("Linear", "velocity", "velocity", 1e-5, 1e-5)
    step = FusedTrainStep(m, tr, opt, n, ["clusters"], vae=vae, seed=3, graph=True, encode_shape=(n, S))
    rngn = np.random.default_rng(0)
    p0 = [p.detach().clone() for p in m.parameters()]
    for it in range(3):
        genes = torch.from_numpy(np.stack([rngn.permutation(G)[:S] for _ in range(n)])).cuda()
        counts = torch.from_numpy(rngn.poisson(1.0, (n, S)).astype(np.float32)).cuda()
        lab = {"clusters": torch.from_numpy(rngn.integers(0, 14, n)).cuda()}
        loss = float(step(condition=lab, counts_subset=counts, genes_subset=genes))
        assert np.isfinite(loss)
        assert torch.equal(step.x1.view(n, 16, 16), vae.encode(counts, genes))
    assert any(not torch.equal(a, b) for a, b in zip(p0, m.parameters()))


def test_fused_train_step_fp16_overflow_skips_adamw_but_not_the_ema_hook():
    """fp16 overflow inside the one-call step: the backward raises the device flag, the AdamW part of the launch is skipped (parameters,
    moments and the step count stay), the EMA action of that step still runs (the reference's hook fires after every batch,
    models.py:83-87), and the next step backs the loss scale off."""
    from scldm_amd.ema import EMA
    from scldm_amd.optim import AdamW
    from scldm_amd.training import FusedTrainStep
    from scldm_amd.transport import create_transport
    vocab = {"cell_line": 4, "gene": 2024}
    n = 64
    m, _, _ = build(vocab, "joint", 8, 85)
    m.precision = "fp16"
    m.cfg_dropout_prob = 0.5
    opt = AdamW([p for p in m.parameters() if p.requires_grad], lr=1e-3)
    ema = EMA(model=m, beta=0.9, update_every=1, update_after_step=0)
    step = FusedTrainStep(m, create_transport("Linear", "velocity", "velocity", 1e-5, 1e-5), opt, n, list(vocab), ema=ema, seed=1, graph=False)
    gen = torch.Generator(device="cuda").manual_seed(3)
    x1 = torch.randn(n, 16, 16, device="cuda", generator=gen)
    cond = {"cell_line": torch.randint(0, 4, (n,), device="cuda", generator=gen), "gene": torch.randint(0, 2024, (n,), device="cuda", generator=gen)}
    step(x1, cond); ema.update()
    assert float(m.found_inf_flag()) == 0.0 and float(opt.param_groups[0]["_step_t"]) == 1.0
    for _ in range(8):          # grow weights (inside the fp16 range themselves) until a product of the backward leaves it
        with torch.no_grad():
            for blk in m.blocks:
                blk.mlp.c_proj.weight.mul_(10.0)
                blk.mlp.w1.weight.mul_(3.0)
        before = {k: p.detach().clone() for k, p in m.named_parameters()}
        ema_before = [p.detach().clone() for p in ema.ema_model.parameters()]
        t_before = float(opt.param_groups[0]["_step_t"])
        step(x1, cond); ema.update()
        if float(m.found_inf_flag()) == 1.0:
            break
    assert float(m.found_inf_flag()) == 1.0
    assert all(torch.equal(before[k], p.detach()) for k, p in m.named_parameters()), "AdamW must skip a step whose gradients are not finite"
    assert float(opt.param_groups[0]["_step_t"]) == t_before
    moved = [not torch.equal(a, b) for a, b in zip(ema_before, ema.ema_model.parameters())]
    assert any(moved), "the EMA hook runs after every batch, also after a skipped optimizer step"
    step(x1, cond); ema.update()
    assert m.fp16_train_state()["headroom"] <= -1


@pytest.mark.parametrize("graph", [True, False])
def test_fused_train_step_follows_a_checkpoint_resume(graph):
    """optimizer.load_state_dict replaces the moment tensors and drops the device launch table: the step's launch struct and captured
    graph hold their addresses (a replay would update freed memory).  FusedTrainStep re-binds itself (AdamW._generation), keeps the
    loaded step count, and a run resumed from a checkpoint after 3 steps ends where the uninterrupted 6-step run ends, bit for bit;
    zero_grad(set_to_none=True) by the caller is survived; GraphedTrainStep refuses to replay after such a load."""
    from scldm_amd.ema import EMA
    from scldm_amd.optim import AdamW
    from scldm_amd.training import FusedTrainStep, GraphedTrainStep
    from scldm_amd.transport import create_transport
    vocab, n = {"cell_line": 4, "gene": 2024}, 64
    m, _, _ = build(vocab, "joint", 8, 47)
    m.precision = "bf16"
    m.cfg_dropout_prob = 0.5
    tr = create_transport("Linear", "velocity", "velocity", 1e-5, 1e-5)
    opt = AdamW([p for p in m.parameters() if p.requires_grad], lr=1e-3, weight_decay=0.01)
    ema = EMA(model=m, beta=0.9, update_every=2, update_after_step=1)
    fs = FusedTrainStep(m, tr, opt, n, list(vocab), ema=ema, seed=5, graph=graph)
    gen = torch.Generator(device="cuda").manual_seed(8)
    batches = [(torch.randn(n, 16, 16, device="cuda", generator=gen), {k: torch.randint(0, v, (n,), device="cuda", generator=gen) for k, v in vocab.items()})
               for _ in range(6)]
    for s in range(3):
        fs(*batches[s]); ema.update()
    ck = copy.deepcopy({"opt": opt.state_dict(), "ema": ema.state_dict(), "rng": fs.rng.clone()})     # (ema's state holds the online model too)
    for s in range(3, 6):
        fs(*batches[s]); ema.update()
    want = [p.detach().clone() for p in m.parameters()], [p.detach().clone() for p in ema.ema_model.parameters()]
    want_m = [opt.state[p]["exp_avg"].clone() for p in opt.param_groups[0]["params"]]
    # resume into the SAME objects (the trainer's flow: build everything, then load the checkpoint)
    ema.load_state_dict(ck["ema"])
    opt.load_state_dict(ck["opt"])
    fs.rng.copy_(ck["rng"])
    opt.zero_grad(set_to_none=True)
    for s in range(3, 6):
        fs(*batches[s]); ema.update()
    assert float(opt.param_groups[0]["_step_t"]) == 6.0
    for a, b in zip(want[0], m.parameters()):
        assert torch.equal(a, b)
    for a, b in zip(want[1], ema.ema_model.parameters()):
        assert torch.equal(a, b)
    for a, p in zip(want_m, opt.param_groups[0]["params"]):
        assert torch.equal(a, opt.state[p]["exp_avg"])
    if graph:
        assert fs.graph is not None
        m2, _, _ = build({"clusters": 14}, "mutually_exclusive", 8, 48)
        m2.precision = "bf16"
        o2 = AdamW([p for p in m2.parameters() if p.requires_grad], lr=1e-3)
        x1, cond = torch.randn(32, 16, 16, device="cuda"), {"clusters": torch.randint(0, 14, (32,), device="cuda")}
        gs = GraphedTrainStep(m2, tr, o2, x1, cond)
        gs(x1, cond)
        o2.load_state_dict(copy.deepcopy(o2.state_dict()))
        with pytest.raises(RuntimeError, match="load checkpoints first"):
            gs(x1, cond)


def test_adamw_gradient_norm_clipping_is_torchs_clip_grad_norm():
    """The trainer's gradient_clip_val (experiments/configs/training/default.yaml:15-16 -> Lightning -> torch.nn.utils.clip_grad_norm_) inside
    the AdamW launch: the norm equals clip_grad_norm_'s return value, the parameters follow torch's clip + fused AdamW over steps whose
    gradient scale varies by 10^4 (Adam alone is scale-invariant: a constant clip would be invisible), `.grad` stays unscaled, a threshold
    above every norm is the unclipped optimizer bit for bit, and a captured graph follows the per-step coefficient."""
    from scldm_amd.optim import AdamW
    torch.manual_seed(1)
    shapes = [(300, 17), (4096,), (5,), (64, 64), (1, 9000), (37,)]
    mk = lambda: torch.nn.ParameterList([torch.nn.Parameter(torch.randn(s, generator=torch.Generator().manual_seed(i)).cuda()) for i, s in enumerate(shapes)])
    net, ref, free, big = mk(), mk(), mk(), mk()
    opt = AdamW(net.parameters(), lr=1e-2, weight_decay=0.05, max_grad_norm=10.0)
    topt = torch.optim.AdamW(ref.parameters(), lr=1e-2, weight_decay=0.05, fused=True)
    fopt = AdamW(free.parameters(), lr=1e-2, weight_decay=0.05)
    bopt = AdamW(big.parameters(), lr=1e-2, weight_decay=0.05, max_grad_norm=1e9)
    scales = [1.0, 100.0, 0.01, 3.0, 0.05, 40.0, 1.0, 0.2]
    clipped = 0
    for step, sc in enumerate(scales):
        grads = [torch.randn(s, generator=torch.Generator().manual_seed(100 * step + i)).cuda() * sc for i, s in enumerate(shapes)]
        for ps in (net, ref, free, big):
            for p, g_ in zip(ps, grads):
                p.grad = g_.clone()
        want_norm = torch.nn.utils.clip_grad_norm_(ref.parameters(), 10.0)
        topt.step()
        opt.step(); fopt.step(); bopt.step()
        got = float(opt.last_grad_norm)
        assert abs(got - float(want_norm)) <= 2e-6 * float(want_norm), (step, got, float(want_norm))
        assert abs(float(opt.param_groups[0]["_clip_ws"][1]) - min(1.0, 10.0 / (float(want_norm) + 1e-6))) < 1e-6
        clipped += float(want_norm) > 10.0
        for p, g_ in zip(net, grads):
            assert torch.equal(p.grad, g_)                                          # the clip is applied in registers, not to .grad
    assert 2 <= clipped <= len(scales) - 2
    for p, q in zip(net, ref):
        assert float((p - q).detach().abs().max()) <= 4e-6 * float(q.detach().abs().max())
    for p, q in zip(big, free):
        assert torch.equal(p, q)                                                    # coefficient 1.0: g * 1.0 is g
    assert any(not torch.equal(p, q) for p, q in zip(net, free))
    with pytest.raises(ValueError):
        AdamW(mk().parameters(), max_grad_norm=0.0)
    two = mk()
    o2 = AdamW([{"params": list(two)[:3]}, {"params": list(two)[3:]}], lr=1e-2, max_grad_norm=1.0)
    for p in two:
        p.grad = torch.ones_like(p)
    with pytest.raises(NotImplementedError, match="one parameter group"):
        o2.step()


def test_fused_train_step_clips_the_gradient_norm_in_the_graph():
    """FusedTrainStep(grad_clip_norm=...): last_grad_norm is the norm of the step's gradients; a threshold above it is the unclipped step
    bit for bit; a threshold far below changes the trajectory, and the HIP-graph replay follows the eager call bit for bit (the coefficient
    is computed on the device per replay)."""
    from scldm_amd.optim import AdamW
    from scldm_amd.training import FusedTrainStep
    from scldm_amd.transport import create_transport
    vocab, n = {"cell_line": 4, "gene": 2024}, 64
    tr = create_transport("Linear", "velocity", "velocity", 1e-5, 1e-5)
    gen = torch.Generator(device="cuda").manual_seed(8)
    batches = [(torch.randn(n, 16, 16, device="cuda", generator=gen) * (1.0 + 3.0 * (s % 2)),
                {k: torch.randint(0, v, (n,), device="cuda", generator=gen) for k, v in vocab.items()}) for s in range(4)]
    m0, _, _ = build(vocab, "joint", 8, 61)
    m0.precision = "bf16"
    m0.cfg_dropout_prob = 0.5

    def run(clip, graph):
        m = copy.deepcopy(m0)
        opt = AdamW([p for p in m.parameters() if p.requires_grad], lr=1e-3, weight_decay=0.01)
        fs = FusedTrainStep(m, tr, opt, n, list(vocab), seed=5, graph=graph, grad_clip_norm=clip)
        norms = []
        for b in batches:
            fs(*b)
            if clip is not None:
                flat = torch.cat([fs.flat] + ([fs._gpos.reshape(-1)] if fs._gpos is not None else []))
                want = float(torch.linalg.vector_norm(flat.double()))
                norms.append((float(opt.last_grad_norm), want))
        return [p.detach().clone() for p in m.parameters()], norms
    free, _ = run(None, False)
    big, norms = run(1e9, False)
    for a, b in zip(free, big):
        assert torch.equal(a, b)
    for got, want in norms:
        assert abs(got - want) <= 2e-6 * want
    thr = 0.5 * min(w for _, w in norms)
    eager, _ = run(thr, False)
    graph, _ = run(thr, True)
    for a, b in zip(eager, graph):
        assert torch.equal(a, b)
    assert any(not torch.equal(a, b) for a, b in zip(eager, free))


def test_fused_train_step_data_parallel_branch_on_a_one_rank_rccl_group():
    """FusedTrainStep(group=...): the call stops after the backward, the flat gradient buffer (and pos_embed's) is all-reduced in place over
    RCCL, then optimizer.step() applies clipping, AdamW and the EMA action.  With a one-rank group the collectives are identities, so the
    result must be the single-process fused step bit for bit - the plumbing (NULL optimizer launch, in-place slices of the flat buffer,
    table built against the views, EMA schedule through step()) is the multi-GPU one; >= 2 ranks never ran on this pool."""
    import os, socket
    import torch.distributed as dist
    from scldm_amd.ema import EMA
    from scldm_amd.optim import AdamW
    from scldm_amd.training import FusedTrainStep
    from scldm_amd.transport import create_transport
    if not dist.is_initialized():
        s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        vocab, n = {"cell_line": 4, "gene": 2024}, 64
        tr = create_transport("Linear", "velocity", "velocity", 1e-5, 1e-5)
        gen = torch.Generator(device="cuda").manual_seed(12)
        batches = [(torch.randn(n, 16, 16, device="cuda", generator=gen), {k: torch.randint(0, v, (n,), device="cuda", generator=gen) for k, v in vocab.items()})
                   for _ in range(4)]
        m0, _, _ = build(vocab, "joint", 8, 71)
        m0.precision = "bf16"
        m0.cfg_dropout_prob = 0.5
        out = {}
        for name, group in (("single", None), ("group", dist.group.WORLD)):
            m = copy.deepcopy(m0)
            opt = AdamW([p for p in m.parameters() if p.requires_grad], lr=1e-3, weight_decay=0.01)
            ema = EMA(model=m, beta=0.9, update_every=2, update_after_step=1)
            fs = FusedTrainStep(m, tr, opt, n, list(vocab), ema=ema, seed=5, graph=False, group=group, grad_clip_norm=0.5)
            assert fs.distributed == (group is not None) and (fs._opt is None) == (group is not None)
            losses = []
            for b in batches:
                losses.append(float(fs(*b)))
                ema.update()
            out[name] = ([p.detach().clone() for p in m.parameters()], [p.detach().clone() for p in ema.ema_model.parameters()], losses,
                         float(opt.last_grad_norm))
        assert out["single"][2] == out["group"][2] and out["single"][3] == out["group"][3]
        for a, b in zip(out["single"][0], out["group"][0]):
            assert torch.equal(a, b)
        for a, b in zip(out["single"][1], out["group"][1]):
            assert torch.equal(a, b)
    finally:
        dist.destroy_process_group()
