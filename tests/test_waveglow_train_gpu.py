"""GPU parity of the WaveGlow training step (forward with saves -> WaveGlowLoss -> hand-written backward):
every parameter gradient against CPU autograd through the oracle, and against the gradients the reference
itself produced (tests/golden/waveglow_small_grads.npz)."""
import os

import numpy as np
import pytest
import torch

from text2speech_amd import _lib, synth

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _rel(a, b):
    a = torch.as_tensor(a).double().cpu().flatten()
    b = torch.as_tensor(b).double().cpu().flatten()
    return float((a - b).norm() / (b.norm() + 1e-30))


def _train_once(cfg, batch, n_samples, seed):
    """One forward+backward on the GPU and the same through the CPU oracle."""
    assert torch.cuda.is_available()
    _lib.load()
    from oracle import waveglow_oracle as O
    from text2speech_amd.glow import WaveGlow, WaveGlowLoss
    sd = synth.waveglow_state(cfg)
    mel, audio = synth.waveglow_inputs(batch, n_samples, seed=seed)
    m = WaveGlow(**cfg)
    m.load_state_dict(sd)
    m = m.to(DEV).train()
    out = m((mel.to(DEV), audio.to(DEV)))
    loss = WaveGlowLoss(1.0)(out)
    loss.backward()
    torch.cuda.synchronize()
    got = {n: p.grad.detach().cpu() for n, p in m.named_parameters() if p.grad is not None}
    # CPU autograd through the oracle (f32)
    sd_cpu = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    outo = O.waveglow_forward(sd_cpu, cfg, mel, audio)
    losso = O.waveglow_loss(outo)
    losso.backward()
    want = {k: v.grad for k, v in sd_cpu.items() if v.grad is not None}
    return dict(got=got, want=want, loss=float(loss), loss_o=float(losso), model=m, n_params=len(list(m.parameters())))


@pytest.fixture(scope="module")
def trained():
    return _train_once(synth.WAVEGLOW_SMALL, 2, 4096, 31)


@pytest.fixture(scope="module")
def trained512():
    """BASELINE configs[3] arithmetic: reference waveglow/config.json defaults (512 channels: 4 M tiles in the gate-backward
    GEMM, 9 x 4-tile weight-gradient GEMMs with flattened split-K, 16 K-chunks in pack_transposed) on a short segment."""
    return _train_once(synth.WAVEGLOW_DEFAULT, 2, 4096, 33)


def test_config_defaults_512ch_loss_and_all_param_grads_vs_oracle(trained512):
    t = trained512
    assert abs(t["loss"] - t["loss_o"]) < 1e-4
    got, want = t["got"], t["want"]
    assert len(got) == t["n_params"] == 938, "a parameter received no gradient"
    worst = []
    for name, w in want.items():
        assert name in got, name
        assert got[name].shape == w.shape, name
        worst.append((_rel(got[name], w), name))
    worst.sort(reverse=True)
    assert worst[0][0] < 2e-3, worst[:8]
    assert sorted(r for r, _ in worst)[len(worst) // 2] < 2e-4, worst[:8]


def test_config_defaults_512ch_grads_vs_reference_golden(trained512, golden_dir):
    """Against the gradients the REFERENCE produced at config.json defaults (tests/golden/waveglow_full_grads.npz): the loss,
    the squared norm of every one of the 938 parameter gradients, and strided samples of one tensor of every kind in flows
    0, 5 and 11."""
    g = np.load(os.path.join(golden_dir, "waveglow_full_grads.npz"))
    got = trained512["got"]
    assert abs(trained512["loss"] - float(g["loss"])) < 1e-4
    names = [str(n) for n in g["all_names"]]
    assert sorted(names) == sorted(got)
    for n, gq in zip(names, g["all_gradsq"]):
        sq = float((got[n].double() ** 2).sum())
        assert abs(sq - gq) <= 5e-3 * gq + 1e-12, (n, sq, gq)
    n_samples = 0
    for key in g.files:
        if not key.startswith("grad::"):
            continue
        name = key[len("grad::"):]
        flat = got[name].flatten()
        step = max(1, flat.numel() // 4096)
        assert _rel(flat[::step], g[key]) < 2e-3, name
        n_samples += 1
    assert n_samples >= 100


def test_benchmarked_shape_8x16000_vs_reference_golden(golden_dir):
    """BASELINE configs[3] at the shape bench.py times (8 x 16000, config.json defaults; split-K of the weight-gradient GEMMs
    = 7 / 21 slabs here against 4 at 2 x 4096): loss, the squared norm and the sum of EVERY one of the 938 parameter gradients,
    and strided samples of one tensor of every kind in flows 0, 5, 11, against the reference's own gradients
    (tests/golden/waveglow_train_full_grads.npz = eight reference backward passes at B = 1, averaged: tools/gen_golden_waveglow.py
    --train-full)."""
    from text2speech_amd.glow import WaveGlow, WaveGlowLoss
    _lib.load()
    g = np.load(os.path.join(golden_dir, "waveglow_train_full_grads.npz"))
    cfg = synth.WAVEGLOW_DEFAULT
    mel, audio = synth.waveglow_inputs(8, 16000, seed=31)
    m = WaveGlow(**cfg)
    m.load_state_dict(synth.waveglow_state(cfg))
    m = m.to(DEV).train()
    loss = WaveGlowLoss(1.0)(m((mel.to(DEV), audio.to(DEV))))
    loss.backward()
    torch.cuda.synchronize()
    assert abs(float(loss) - float(g["loss"])) < 1e-4, (float(loss), float(g["loss"]))
    got = {n: p.grad.detach() for n, p in m.named_parameters() if p.grad is not None}
    names = [str(n) for n in g["all_names"]]
    assert sorted(names) == sorted(got) and len(names) == 938
    worst_sq = 0.0
    for n, gq, gs in zip(names, g["all_gradsq"], g["all_gradsum"]):
        gr = got[n].double()
        sq = float((gr ** 2).sum())
        worst_sq = max(worst_sq, abs(sq - gq) / gq)
        assert abs(sq - gq) <= 5e-3 * gq + 1e-14, (n, sq, gq)
        # the plain sum cancels: hold it to 2e-3 of the tensor's 1-norm scale sqrt(numel * sum of squares)
        assert abs(float(gr.sum()) - gs) <= 2e-3 * (gr.numel() * gq) ** 0.5 + 1e-12, (n, float(gr.sum()), gs)
    worst = 0.0
    n_samples = 0
    for key in g.files:
        if not key.startswith("grad::"):
            continue
        name = key[len("grad::"):]
        flat = got[name].flatten()
        step = max(1, flat.numel() // 2048)
        r = _rel(flat[::step], g[key])
        worst = max(worst, r)
        assert r < 2e-3, (name, r)
        n_samples += 1
    assert n_samples >= 100
    print("8x16000 train step vs reference: worst |dsq|/sq %.2e, worst sampled rel %.2e" % (worst_sq, worst))


def test_four_layer_wn_grads_vs_oracle():
    """n_layers = 4: the largest dilated tap offset is 8 rows, less than the 32-row K-block of the channel-last weight-gradient
    GEMM - the plane margin is rounded up to 32 rows (glow._Engine.geom) so that its shifted K-blocks stay inside the planes."""
    cfg = dict(synth.WAVEGLOW_SMALL, WN_config=dict(n_layers=4, n_channels=64, kernel_size=3))
    t = _train_once(cfg, 2, 2048, 35)
    assert t["model"]._eng().geom()["halo"] == 32
    assert abs(t["loss"] - t["loss_o"]) < 1e-4
    worst = sorted(((_rel(t["got"][n], w), n) for n, w in t["want"].items()), reverse=True)
    assert len(t["got"]) == t["n_params"] and worst[0][0] < 2e-3, worst[:6]


def test_config_defaults_512ch_backward_is_bitwise_reproducible(trained512):
    from text2speech_amd.glow import WaveGlowLoss
    m = trained512["model"]
    mel, audio = synth.waveglow_inputs(2, 4096, seed=33)
    for _ in range(2):
        m.zero_grad(set_to_none=True)
        WaveGlowLoss(1.0)(m((mel.to(DEV), audio.to(DEV)))).backward()
        torch.cuda.synchronize()
        for n, p in m.named_parameters():
            assert torch.equal(p.grad.cpu(), trained512["got"][n]), n


def test_loss_and_all_param_grads_vs_oracle(trained):
    assert abs(trained["loss"] - trained["loss_o"]) < 1e-4
    got, want = trained["got"], trained["want"]
    assert len(got) == trained["n_params"], "a parameter received no gradient"
    worst = []
    for name, w in want.items():
        assert name in got, name
        assert got[name].shape == w.shape, name
        worst.append((_rel(got[name], w), name))
    worst.sort(reverse=True)
    assert worst[0][0] < 2e-3, worst[:8]
    # the bulk is far tighter than the bar
    assert sorted(r for r, _ in worst)[len(worst) // 2] < 2e-4, worst[:8]


def test_grads_vs_reference_golden(trained, golden_dir):
    g = np.load(os.path.join(golden_dir, "waveglow_small_grads.npz"))
    assert abs(trained["loss"] - float(g["loss"])) < 1e-4
    for key in g.files:
        if not key.startswith("grad::"):
            continue
        name = key[len("grad::"):]
        flat = trained["got"][name].flatten()
        step = max(1, flat.numel() // 32768)
        assert _rel(flat[::step], g[key]) < 2e-3, name
        sq = float((flat.double() ** 2).sum())
        assert abs(sq - float(g["gradsq::" + name])) < 5e-3 * float(g["gradsq::" + name]), name


def test_adam_step_matches_torch(trained):
    """FusedAdam (one table-driven launch) vs torch.optim.Adam on the same gradients."""
    from text2speech_amd.optim import FusedAdam
    m = trained["model"]
    ref = [p.detach().clone().cpu().requires_grad_(True) for p in m.parameters()]
    for r, p in zip(ref, m.parameters()):
        r.grad = p.grad.detach().cpu().clone()
    opt_ref = torch.optim.Adam(ref, lr=1e-4)
    opt = FusedAdam(m.parameters(), lr=1e-4)
    for _ in range(3):
        opt.step()
        opt_ref.step()
    torch.cuda.synchronize()
    worst = max(_rel(p.detach(), r.detach()) for p, r in zip(m.parameters(), ref))
    assert worst < 1e-6, worst


def test_adam_state_interop_and_cache_invalidation():
    """(1) an optimizer state written by torch.optim.Adam (what the reference's checkpoints hold, waveglow/train.py:41-60)
    loads into FusedAdam and continues with the right bias correction; (2) load_state_dict on an optimizer that has already
    stepped re-targets the kernel's job table at the new moment buffers; (3) a step invalidates the packed-weight cache that
    infer() keys on parameter versions, so infer() after step() uses the new weights."""
    from text2speech_amd.glow import WaveGlow, WaveGlowLoss
    from text2speech_amd.optim import FusedAdam
    _lib.load()
    cfg = synth.WAVEGLOW_SMALL
    mel, audio = synth.waveglow_inputs(1, 2048, seed=3)
    m = WaveGlow(**cfg)
    m.load_state_dict(synth.waveglow_state(cfg))
    m = m.to(DEV).train()
    WaveGlowLoss(1.0)(m((mel.to(DEV), audio.to(DEV)))).backward()
    torch.cuda.synchronize()
    params = list(m.parameters())
    ref = [p.detach().clone().requires_grad_(True) for p in params]
    for r, p in zip(ref, params):
        r.grad = p.grad.detach().clone()
    opt_ref = torch.optim.Adam(ref, lr=1e-3)
    for _ in range(5):
        opt_ref.step()
    opt = FusedAdam(params, lr=1e-3)
    opt.step()                                   # has stepped once: job table cached, moments allocated
    with torch.no_grad():
        for p, r in zip(params, ref):
            p.copy_(r)
    import copy
    # torch.optim.Adam layout: per-parameter 'step' tensors.  deepcopy: Optimizer.load_state_dict keeps tensors that already
    # have the parameter's dtype and device BY REFERENCE, and the two optimizers must not share moment buffers here
    opt.load_state_dict(copy.deepcopy(opt_ref.state_dict()))
    opt.step()
    opt_ref.step()
    torch.cuda.synchronize()
    assert opt.param_groups[0]["step"] == 6
    worst = max(_rel(p.detach(), r.detach()) for p, r in zip(params, ref))
    assert worst < 1e-6, worst
    for p, r in zip(params[:20], ref[:20]):      # the moments the kernel wrote are the loaded ones, advanced by one step
        assert _rel(opt.state[p]["exp_avg"], opt_ref.state[r]["exp_avg"]) < 1e-5
        assert _rel(opt.state[p]["exp_avg_sq"], opt_ref.state[r]["exp_avg_sq"]) < 1e-5
    # and the state written here loads into torch.optim.Adam
    back = torch.optim.Adam(ref, lr=1e-3)
    back.load_state_dict(copy.deepcopy(opt.state_dict()))
    assert int(back.state[ref[0]]["step"]) == 6
    # ... and CONTINUES there: every parameter has a step tensor of its own (a shared one would advance once per parameter)
    steps = [back.state[r]["step"] for r in ref]
    assert len({id(s) for s in steps}) == len(steps)
    before = [r.detach().clone() for r in ref]
    back.step()
    opt.step()
    torch.cuda.synchronize()
    assert all(int(back.state[r]["step"]) == 7 for r in ref) and opt.param_groups[0]["step"] == 7
    assert any(not torch.equal(b, r.detach()) for b, r in zip(before, ref))
    worst = max(_rel(p.detach(), r.detach()) for p, r in zip(params, ref))
    assert worst < 1e-6, worst
    # (3) infer() after a step must see the stepped weights
    m.eval()
    gen = torch.Generator().manual_seed(1)
    spect = torch.randn(1, 80, 6, generator=gen).to(DEV)
    L = 6 * 256 // 8
    noise = (torch.randn(1, 4, L, generator=gen), [torch.randn(1, 2, L, generator=gen) for _ in range(2)])
    a0 = m.infer(spect, sigma=0.7, noise=noise).clone()
    opt2 = FusedAdam(params, lr=1e-2)
    opt2.step()
    a1 = m.infer(spect, sigma=0.7, noise=noise).clone()
    m2 = WaveGlow(**cfg)
    m2.load_state_dict(m.state_dict())
    a2 = m2.to(DEV).eval().infer(spect, sigma=0.7, noise=noise)
    assert not torch.equal(a0, a1)
    assert torch.equal(a1, a2), _rel(a1, a2)


def test_bucketed_allreduce_inside_backward_nccl_world1():
    """The DP path end to end on one GPU: RCCL process group of size 1, weights broadcast, 13 flat gradient
    buckets all-reduced from inside the backward; gradients must equal the single-process ones."""
    import torch.distributed as dist
    from text2speech_amd import distributed as D
    from text2speech_amd.glow import WaveGlow, WaveGlowLoss
    cfg = synth.WAVEGLOW_SMALL
    sd = synth.waveglow_state(cfg)
    mel, audio = synth.waveglow_inputs(1, 2048, seed=9)

    def run(with_dp):
        m = WaveGlow(**cfg)
        m.load_state_dict(sd)
        m = m.to(DEV).train()
        if with_dp:
            D.apply_gradient_allreduce(m)
        WaveGlowLoss(1.0)(m((mel.to(DEV), audio.to(DEV)))).backward()
        torch.cuda.synchronize()
        return m, {n: p.grad.detach().clone() for n, p in m.named_parameters()}

    _, base = run(False)
    D.init_distributed(0, 1, None, "nccl", "tcp://127.0.0.1:29517")
    try:
        m, got = run(True)
        sync = m._eng().grad_sync
        assert sync is not None and sync.n_buckets == cfg["n_flows"] + 1 and not sync.pending
        for n in base:
            assert torch.equal(base[n], got[n]), n
    finally:
        dist.destroy_process_group()


def test_backward_is_bitwise_reproducible():
    """The backward runs on two HIP streams with events on the shared d-plane buffers and uses no atomics: the same step twice
    must give bit-identical gradients (a missing dependency between the streams would show up here as run-to-run noise)."""
    from text2speech_amd.glow import WaveGlow, WaveGlowLoss
    _lib.load()
    cfg = synth.WAVEGLOW_SMALL
    mel, audio = synth.waveglow_inputs(2, 4096, seed=32)
    m = WaveGlow(**cfg)
    m.load_state_dict(synth.waveglow_state(cfg))
    m = m.to(DEV).train()
    runs = []
    for _ in range(3):
        m.zero_grad(set_to_none=True)
        WaveGlowLoss(1.0)(m((mel.to(DEV), audio.to(DEV)))).backward()
        torch.cuda.synchronize()
        runs.append({n: p.grad.detach().clone() for n, p in m.named_parameters() if p.grad is not None})
    for n in runs[0]:
        assert torch.equal(runs[0][n], runs[1][n]) and torch.equal(runs[0][n], runs[2][n]), n
