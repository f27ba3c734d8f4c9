"""GPU: the drop-in modules and the full GAN train step (all through libeben_hip.so) against the
CPU oracle and the golden fixtures frozen from the reference.

Gradient comparisons use the relative L2 norm per tensor: the graph is discontinuous (LeakyReLU
masks, sign(a-b) of the L1 feature loss) and fp32 rounding noise flips isolated elements -- the
reference's own fp32 and fp64 gradients differ by `check:disc_grad_fp64_floor` (~1e-2) in that
metric on the discriminator, which is the bar used here.
"""
from functools import partial

import numpy as np
import pytest
import torch

from formula import formula_audio, formula_state_dict, iter_embeddings, summarize
from oracle import eben_oracle as O

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda")


def rel_l2(got, ref):
    got, ref = got.detach().double().cpu(), ref.detach().double().cpu()
    assert got.shape == ref.shape and torch.isfinite(got).all()
    return float((got - ref).norm() / (ref.norm() + 1e-30))


def max_abs(got, ref):
    return float((got.detach().double().cpu() - ref.detach().double().cpu()).abs().max())


def contract(golden, tag):
    return {k: tuple(int(x) for x in s.split(",")) for k, s in zip(golden[f"contract/{tag}/keys"], golden[f"contract/{tag}/shapes"])}


def build_generator(golden, p):
    from vibravox_amd.torch_modules.dnn.eben_generator import EBENGenerator

    gen = EBENGenerator(m=4, n=32, p=p)
    shapes = contract(golden, "G")
    shapes["first_conv.weight"] = (32, p, 3)
    assert list(gen.state_dict().keys()) == list(contract(golden, "G").keys())
    sd = formula_state_dict(shapes, f"G{p}")
    missing, unexpected = gen.load_state_dict(sd, strict=False)
    assert not unexpected and all(k.startswith("pqmf.") for k in missing)
    np.testing.assert_array_equal(gen.pqmf.analysis_weights.numpy(), golden["pqmf/analysis_4_32"])
    osd = {k: v.detach().clone() for k, v in gen.state_dict().items()}
    return gen.to(DEV), osd


def build_discriminator(golden):
    from vibravox_amd.torch_modules.dnn.eben_discriminator import DiscriminatorEBENMultiScales

    disc = DiscriminatorEBENMultiScales(q=4, min_channels=24)
    shapes = contract(golden, "D")
    assert {k: tuple(v.shape) for k, v in disc.state_dict().items()} == shapes
    sd = formula_state_dict(shapes, "D")
    disc.load_state_dict(sd, strict=True)
    return disc.to(DEV), {k: v.clone() for k, v in sd.items()}


def check_summary(golden, prefix, t, rtol=5e-5, atol=5e-6):
    s = summarize(t.detach().cpu())
    assert tuple(golden[f"{prefix}:shape"]) == tuple(s["shape"])
    np.testing.assert_allclose(s["probe"], golden[f"{prefix}:probe"], rtol=rtol, atol=atol)
    np.testing.assert_allclose(s["l2"], golden[f"{prefix}:l2"], rtol=rtol)


@pytest.mark.parametrize("p", [2, 1])
def test_generator_matches_oracle_and_reference_golden(hip, golden, p):
    gen, osd = build_generator(golden, p)
    x = O.cut_to_valid_length(formula_audio("g_in", 2, 8192))
    enh, bands = gen(x.to(DEV))
    assert enh.shape == x.shape  # the reference's own test (tests/torch_modules/eben_generator_test.py:2-8)
    check_summary(golden, f"gen{p}/enhanced", enh)
    check_summary(golden, f"gen{p}/bands", bands)
    osd = {k: v.requires_grad_(not k.startswith("pqmf.")) for k, v in osd.items()}
    o_enh, o_bands = O.generator_forward(osd, x, p)
    assert max_abs(enh, o_enh) < 2e-5 and max_abs(bands, o_bands) < 2e-5
    mse = float(((enh.detach().cpu().double() - o_enh.detach().double()) ** 2).mean())
    assert mse < 1e-10  # north_star bar is 1e-5; fp32 floor ~5e-15
    wgt = formula_audio("g_seed", 2, enh.shape[2], amp=1.0)
    ((enh * wgt.to(DEV)).sum() + (bands ** 2).sum()).backward()
    ((o_enh * wgt).sum() + (o_bands ** 2).sum()).backward()
    worst = 0.0
    for k, prm in gen.named_parameters():
        if prm.grad is None:
            continue
        worst = max(worst, rel_l2(prm.grad, osd[k].grad))
        np.testing.assert_allclose(prm.grad.double().norm().item(), golden[f"gen{p}/grad_l2/{k}"], rtol=2e-3)
    assert worst < 2e-3, worst


def test_generator_config1_forward(hip, golden):
    """BASELINE config 1: generator-only forward, batch 4 x 16000 -> 15840."""
    gen, _ = build_generator(golden, 2)
    x = gen.cut_to_valid_length(formula_audio("cfg1", 4, 16000))
    assert x.shape[2] == 15840
    with torch.no_grad():
        enh, bands = gen(x.to(DEV))
    check_summary(golden, "cfg1/enhanced", enh)
    check_summary(golden, "cfg1/bands", bands)


def test_discriminator_losses_grads(hip, golden):
    from vibravox_amd.torch_modules.losses.feature_loss import FeatureLossForDiscriminatorMelganMultiScales
    from vibravox_amd.torch_modules.losses.hinge_loss import HingeLossForDiscriminatorMelganMultiScales

    disc, osd = build_discriminator(golden)
    bands = formula_audio("d_bands", 8, 2016, amp=0.5).reshape(2, 4, 2016)
    audio = formula_audio("d_audio", 2, 4 * 2016 - 32)
    bands_b = formula_audio("d_bands_b", 8, 2016, amp=0.5).reshape(2, 4, 2016)
    audio_b = formula_audio("d_audio_b", 2, 4 * 2016 - 32)
    bd, ad = bands.to(DEV).requires_grad_(True), audio.to(DEV).requires_grad_(True)
    e_a = disc(bands=bd, audio=ad)
    with torch.no_grad():
        e_b = disc(bands=bands_b.to(DEV), audio=audio_b.to(DEV))
    assert [len(s) for s in e_a] == [9, 9, 9, 8]
    for name, t in iter_embeddings(e_a):
        check_summary(golden, f"disc/{name}", t)
    fm, hinge = FeatureLossForDiscriminatorMelganMultiScales(), HingeLossForDiscriminatorMelganMultiScales()
    l_fm, l_hp, l_hm = fm(e_a, e_b), hinge(embeddings=e_a, target=1), hinge(embeddings=e_a, target=-1)
    np.testing.assert_allclose(l_fm.item(), golden["loss/fm"], rtol=2e-5)
    np.testing.assert_allclose(l_hp.item(), golden["loss/hinge_p1"], rtol=1e-5)
    np.testing.assert_allclose(l_hm.item(), golden["loss/hinge_m1"], rtol=1e-5)
    (l_fm + 0.5 * l_hp + 0.25 * l_hm).backward()
    # oracle on the same inputs
    osd = {k: v.requires_grad_(True) for k, v in osd.items()}
    ob, oa = bands.clone().requires_grad_(True), audio.clone().requires_grad_(True)
    o_a = O.discriminator_forward(osd, ob, oa, 4)
    with torch.no_grad():
        o_b = O.discriminator_forward(osd, bands_b, audio_b, 4)
    (O.feature_loss(o_a, o_b) + 0.5 * O.hinge_loss(o_a, 1) + 0.25 * O.hinge_loss(o_a, -1)).backward()
    floor = float(golden["check:disc_grad_fp64_floor"])
    worst = max(rel_l2(bd.grad, ob.grad), rel_l2(ad.grad, oa.grad))
    for k, prm in disc.named_parameters():
        worst = max(worst, rel_l2(prm.grad, osd[k].grad))
        np.testing.assert_allclose(prm.grad.double().norm().item(), golden[f"disc/grad_l2/{k}"], rtol=2e-2)
    assert worst <= 2 * floor + 1e-3, (worst, floor)
    np.testing.assert_allclose(bd.grad.double().norm().item(), golden["disc/grad_bands:l2"], rtol=2e-3)
    np.testing.assert_allclose(ad.grad.double().norm().item(), golden["disc/grad_audio:l2"], rtol=2e-3)


def make_module(golden, use_mrstft, fused_adam=True):
    from vibravox_amd.lightning_modules.eben import EBENLightningModule
    from vibravox_amd.optim import FusedAdam
    from vibravox_amd.torch_modules.losses.feature_loss import FeatureLossForDiscriminatorMelganMultiScales
    from vibravox_amd.torch_modules.losses.hinge_loss import HingeLossForDiscriminatorMelganMultiScales
    from vibravox_amd.torch_modules.losses.mrstft_loss import MultiResolutionSTFTLoss

    gen, g_sd = build_generator(golden, 2)
    disc, d_sd = build_discriminator(golden)
    opt = partial(FusedAdam if fused_adam else torch.optim.Adam, lr=3e-4, betas=(0.5, 0.9))
    mr = MultiResolutionSTFTLoss(fft_sizes=(512, 1024, 2048), hop_sizes=(50, 120, 240), win_lengths=(240, 600, 1200),
                                 sample_rate=16000, perceptual_weighting=True).to(DEV) if use_mrstft else None
    mod = EBENLightningModule(sample_rate=16000, generator=gen, discriminator=disc, generator_optimizer=opt,
                              discriminator_optimizer=opt, reconstructive_loss_freq_fn=mr,
                              feature_matching_loss_fn=FeatureLossForDiscriminatorMelganMultiScales(),
                              adversarial_loss_fn=HingeLossForDiscriminatorMelganMultiScales(),
                              dynamic_loss_balancing="ema", beta_ema=0.9, update_discriminator_ratio=1)
    return mod, g_sd, d_sd


@pytest.mark.parametrize("fused_adam,literal,engine", [(True, False, True), (False, False, True), (True, False, False), (True, True, False)])
def test_two_train_steps_against_reference_replay_golden(hip, golden, fused_adam, literal, engine):
    """eben.py:82-130 replayed over the REFERENCE modules (golden) vs this build's LightningModule:
    the literal as-executed order, the order with the redundant discriminator passes removed (autograd),
    and the batched discriminator engine (one forward, one stacked backward)."""
    mod, g_sd, d_sd = make_module(golden, use_mrstft=False, fused_adam=fused_adam)
    mod.exploit_step_redundancy = not literal
    mod.use_disc_engine = engine
    for i in range(2):
        batch = {"audio_body_conducted": formula_audio(f"step{i}/bc", 2, 8200).to(DEV),
                 "audio_airborne": formula_audio(f"step{i}/air", 2, 8200).to(DEV)}
        out = mod.training_step(batch)
        assert set(out) == {"corrupted", "enhanced", "reference"}
        check_summary(golden, f"step{i}/enhanced", out["enhanced"], rtol=2e-4, atol=2e-5)
        for k in ("train/generator/feature_matching_loss", "train/generator/adv_loss_gen", "train/generator/backprop_loss",
                  "train/discriminator/real_loss", "train/discriminator/fake_loss", "train/discriminator/backprop_loss"):
            np.testing.assert_allclose(mod.logged[k].item(), golden[f"step{i}/{k}"], rtol=5e-4)
        np.testing.assert_allclose(torch.stack(mod.last_norms).cpu().numpy(), golden[f"step{i}/balancing/norms"], rtol=2e-3)
        np.testing.assert_allclose(torch.stack(mod.last_lambdas).cpu().numpy(), golden[f"step{i}/balancing/lambdas"], rtol=2e-3)
    for k, v in mod.generator.state_dict().items():
        if not k.startswith("pqmf."):
            np.testing.assert_allclose(v.double().norm().item(), golden[f"post/G/{k}"][1], rtol=2e-4)
    for k, v in mod.discriminator.state_dict().items():
        np.testing.assert_allclose(v.double().norm().item(), golden[f"post/D/{k}"][1], rtol=2e-4)


@pytest.mark.parametrize("literal,engine", [(False, True), (False, False), (True, False)])
def test_train_step_with_mrstft_against_oracle(hip, golden, literal, engine):
    """Full default configuration (MRSTFT + FM + hinge, EMA balancing).  The MRSTFT term is a
    restatement of third-party auraloss (parity unpinned); everything else is pinned."""
    mod, g_sd, d_sd = make_module(golden, use_mrstft=True)
    mod.exploit_step_redundancy = not literal
    mod.use_disc_engine = engine
    trainer = O.OracleTrainer(g_sd, d_sd, p=2, q=4, use_mrstft=True)
    for i in range(2):
        bc, air = formula_audio(f"mstep{i}/bc", 2, 8200), formula_audio(f"mstep{i}/air", 2, 8200)
        mod.training_step({"audio_body_conducted": bc.to(DEV), "audio_airborne": air.to(DEV)})
        logs = trainer.step(bc, air)
        for k in ("train/generator/reconstructive_loss_freq", "train/generator/feature_matching_loss", "train/generator/adv_loss_gen",
                  "train/generator/backprop_loss", "train/discriminator/real_loss", "train/discriminator/fake_loss"):
            np.testing.assert_allclose(mod.logged[k].item(), logs[k].item(), rtol=1e-3, err_msg=k)
        np.testing.assert_allclose(torch.stack(mod.last_norms).cpu().numpy(), logs["balancing/norms"].numpy(), rtol=5e-3)


@pytest.mark.parametrize("batch,length", [(1, 16000), (1, 4321), (3, 1000)])
def test_generator_inference_variable_length(hip, golden, batch, length):
    """the evaluation path (eben.py:132-165: cut_to_valid_length, generator forward under
    no_grad, any clip length, batch 1) against the oracle on the same weights."""
    gen, osd = build_generator(golden, 2)
    x = formula_audio(f"infer/{batch}/{length}", batch, length)
    with torch.no_grad():
        cut = gen.cut_to_valid_length(x.to(DEV))
        enh, bands = gen(cut)
    o_cut = O.cut_to_valid_length(x)
    assert cut.shape == o_cut.shape and enh.shape == cut.shape
    o_enh, o_bands = O.generator_forward(osd, o_cut, 2)
    assert max_abs(enh, o_enh) < 2e-5 and max_abs(bands, o_bands) < 2e-5
    assert float(((enh.cpu().double() - o_enh.detach().double()) ** 2).mean()) < 1e-10   # north-star bar: MSE < 1e-5


def test_eval_step_logs_reference_losses(hip, golden):
    """eben.py:132-165: validation / test step -- outputs and the logged atomic losses of both networks
    equal the oracle's on the same weights; no gradient state is touched."""
    mod, g_sd, d_sd = make_module(golden, use_mrstft=False)
    bc, air = formula_audio("eval/bc", 2, 8200), formula_audio("eval/air", 2, 8200)
    out = mod.validation_step({"audio_body_conducted": bc.to(DEV), "audio_airborne": air.to(DEV)}, 0)
    assert set(out) == {"corrupted", "enhanced", "reference"} and not out["enhanced"].requires_grad
    assert all(p.grad is None for p in mod.parameters())
    x = O.cut_to_valid_length(bc)
    ref = O.cut_to_valid_length(air)
    o_enh, o_bands = O.generator_forward(g_sd, x, 2)
    assert max_abs(out["enhanced"], o_enh) < 2e-5
    o_ref_bands = O.pqmf_analysis(ref, g_sd["pqmf.analysis_weights"], bands=4)
    e_emb = O.discriminator_forward(d_sd, o_bands, o_enh, 4)
    r_emb = O.discriminator_forward(d_sd, o_ref_bands, ref, 4)
    want = {"validation/generator/feature_matching_loss": O.feature_loss(e_emb, r_emb),
            "validation/generator/adv_loss_gen": O.hinge_loss(e_emb, 1),
            "validation/discriminator/real_loss": O.hinge_loss(r_emb, 1),
            "validation/discriminator/fake_loss": O.hinge_loss(e_emb, -1)}
    for k, v in want.items():
        np.testing.assert_allclose(mod.logged[k].item(), float(v), rtol=5e-4, err_msg=k)
    only_bc = mod.test_step({"audio_body_conducted": bc.to(DEV)}, 0)
    assert set(only_bc) == {"corrupted", "enhanced"}


def test_full_size_step_properties(hip, golden):
    """BASELINE config 2 shape (batch 32 x 32000 -> 31968): size-independent properties only."""
    mod, _, _ = make_module(golden, use_mrstft=True)
    g = torch.Generator().manual_seed(1234)
    batch = {"audio_body_conducted": (0.1 * torch.randn(32, 1, 32000, generator=g)).to(DEV),
             "audio_airborne": (0.1 * torch.randn(32, 1, 32000, generator=g)).to(DEV)}
    before = {k: v.clone() for k, v in mod.discriminator.state_dict().items()}
    out = mod.training_step(batch)
    torch.cuda.synchronize()
    assert out["enhanced"].shape == (32, 1, 31968) and torch.isfinite(out["enhanced"]).all()
    assert out["enhanced"].abs().max() <= 4.0 * 1.05  # tanh-bounded bands through a gain-4 synthesis bank
    for k, v in mod.logged.items():
        assert torch.isfinite(v).all(), k
    lam = torch.stack(mod.last_lambdas)
    assert (lam > 0).all() and (lam <= 1e4).all()
    moved = {k: float((v - before[k]).abs().max()) for k, v in mod.discriminator.state_dict().items()}
    assert max(moved.values()) <= 3e-4 * 1.01  # first Adam step moves every element by <= lr
    # the logit biases see d(real)/db = -1/4 and d(fake)/db = +1/4 while every hinge term is active
    # (near-zero logits at init): exactly zero gradient, so they may stay put; everything else moves
    still = [k for k, m in moved.items() if m == 0.0]
    assert all(k.endswith(".bias") for k in still) and len(still) <= 4, still


def test_bf16_discriminator_math_against_fp32_step(hip, golden):
    """disc_math = "bf16" (BASELINE config 2): the discriminator contractions take bf16 MFMA operands, everything
    else -- the generator, accumulation, losses, Adam -- is unchanged fp32.  One step from the same state in both
    modes: the generator output is IDENTICAL (it never sees the discriminator), the logged losses and balancing
    norms move by ~1e-3, the generator gradient by < 2 % (whole-vector relative L2; Adam's first moment after one
    step is (1 - beta1) * grad).  The discriminator gradient is the difference of two coherent sums (fake and real
    hinge branches, opposite signs) and inherits the bf16 noise of each un-cancelled: tens of percent at
    initialisation -- the stated cost of this mode, measured here and bounded loosely."""
    res = {}
    for math in ("f32", "bf16", "bf16+gen"):
        mod, _, _ = make_module(golden, use_mrstft=True)
        mod.disc_math = math.split("+")[0]
        mod.gen_backward_math = "bf16" if math.endswith("+gen") else "f32"
        mod.stft_math = "bf16x3" if math.endswith("+gen") else None   # bench.py's bf16 step: MRSTFT contractions on hi/lo bf16 splits
        batch = {"audio_body_conducted": formula_audio("bf/bc", 4, 8200).to(DEV), "audio_airborne": formula_audio("bf/air", 4, 8200).to(DEV)}
        out = mod.training_step(batch)
        torch.cuda.synchronize()
        moments = []
        for opt in mod.optimizers():
            moments.append(torch.cat([opt.state[p]["exp_avg"].double().flatten().cpu() for grp in opt.param_groups for p in grp["params"]
                                      if "exp_avg" in opt.state.get(p, {})]))
        res[math] = (out["enhanced"].clone(), {k: float(v) for k, v in mod.logged.items()}, torch.stack(mod.last_norms).cpu(), moments)
    # ... and with the generator's BACKWARD contractions in bf16 as well (bench.py's default): its forward is still exact
    a, c = res["f32"], res["bf16+gen"]
    assert torch.equal(a[0], c[0])
    g_rel = float((a[3][0] - c[3][0]).norm() / a[3][0].norm())
    n_rel = float(((a[2] - c[2]).abs() / a[2].abs()).max())
    print(f"bf16 discriminator + generator-backward math: norms {n_rel:.3e}, generator grad rel-L2 {g_rel:.3e}")
    assert n_rel < 5e-2 and g_rel < 5e-2
    a, b = res["f32"], res["bf16"]
    assert torch.equal(a[0], b[0])
    rel = {k: abs(b[1][k] - v) / abs(v) for k, v in a[1].items()}
    n_rel = float(((a[2] - b[2]).abs() / a[2].abs()).max())
    g_rel = float((a[3][0] - b[3][0]).norm() / a[3][0].norm())
    d_rel = float((a[3][1] - b[3][1]).norm() / a[3][1].norm())
    print(f"bf16 discriminator math: losses {rel}, norms {n_rel:.3e}, generator grad rel-L2 {g_rel:.3e}, discriminator grad rel-L2 {d_rel:.3e}")
    # the feature-matching loss is a sum of L1 distances between nearly equal embeddings (formula weights): percent level
    for k, r in rel.items():
        assert r < (5e-2 if "feature_matching" in k or "backprop" in k else 5e-3), (k, r)
    assert n_rel < 5e-2 and g_rel < 5e-2 and d_rel < 1.0


def test_train_step_is_bitwise_reproducible(hip, golden):
    """Every kernel on the path reduces in a fixed order (no float atomics), so two runs from the same
    state must agree bit for bit -- parameters and Adam moments; a difference is a race between the
    streams the step runs on (main, weight-gradient side stream, the discriminator chains)."""
    def run():
        mod, _, _ = make_module(golden, use_mrstft=True)
        g = torch.Generator().manual_seed(99)
        batch = {"audio_body_conducted": (0.1 * torch.randn(8, 1, 16000, generator=g)).to(DEV),
                 "audio_airborne": (0.1 * torch.randn(8, 1, 16000, generator=g)).to(DEV)}
        for _ in range(2):
            mod.training_step(batch)
        torch.cuda.synchronize()
        out = {f"G.{k}": v.clone() for k, v in mod.generator.state_dict().items()}
        out.update({f"D.{k}": v.clone() for k, v in mod.discriminator.state_dict().items()})
        for oi, opt in enumerate(mod.optimizers()):
            for pi, prm in enumerate(p for grp in opt.param_groups for p in grp["params"]):
                if "exp_avg" in opt.state.get(prm, {}):   # the frozen PQMF banks never get a moment
                    out[f"adam{oi}.m.{pi}"] = opt.state[prm]["exp_avg"].clone()
        return out

    a, b = run(), run()
    assert [k for k in a if not torch.equal(a[k], b[k])] == []


def test_bucketed_grad_sync_path_on_gpu_matches_plain_step(hip, golden):
    """The data-parallel plumbing of bench.py (gradients as views of flat buckets, all-reduce issued
    from post-accumulate hooks on a side stream, 1/N folded into Adam) on a single-rank RCCL group:
    must reproduce the plain single-GPU step."""
    import os

    import torch.distributed as dist

    from vibravox_amd.ddp import BucketedZeroGrad, GradSync

    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29541")
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        plain, _, _ = make_module(golden, use_mrstft=False)
        synced, _, _ = make_module(golden, use_mrstft=False)
        g_opt, d_opt = synced.optimizers()
        gs = GradSync(synced.generator.parameters(), bucket_bytes=1 << 20)
        ds = GradSync(synced.discriminator.parameters(), bucket_bytes=8 << 20)
        g_w, d_w = BucketedZeroGrad(g_opt, gs), BucketedZeroGrad(d_opt, ds)
        synced._optimizers = [g_w, d_w]
        synced.grad_sync = {id(g_w): gs, id(d_w): ds}
        for i in range(2):
            batch = {"audio_body_conducted": formula_audio(f"step{i}/bc", 2, 8200).to(DEV),
                     "audio_airborne": formula_audio(f"step{i}/air", 2, 8200).to(DEV)}
            plain.training_step(batch)
            synced.training_step(batch)
        torch.cuda.synchronize()
        assert gs.launched >= 2 * len(gs.buckets) and ds.launched >= 2 * len(ds.buckets)
        for (k, a), (_, b) in zip(plain.generator.state_dict().items(), synced.generator.state_dict().items()):
            assert torch.allclose(a, b, rtol=0, atol=1e-6), k
        for (k, a), (_, b) in zip(plain.discriminator.state_dict().items(), synced.discriminator.state_dict().items()):
            assert torch.allclose(a, b, rtol=0, atol=1e-6), k
    finally:
        dist.destroy_process_group()
