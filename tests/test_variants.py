"""The reference's OTHER legal ``EBENLightningModule`` configurations (``vibravox/lightning_modules/eben.py:67-76, 118, 194-211,
229-235``) against ``tests/golden/eben_variants_golden.npz`` -- the reference modules replayed by
``tests/golden/make_variants_golden.py``: no balancing, "simple" balancing, ``update_discriminator_ratio`` 0 and 0.5 (the same
``torch.rand(1)`` sequence), an L1 time-domain loss, feature-matching-only, adversarial-only.

CPU: the oracle's ``OracleTrainer`` is pinned to the fixture.  GPU: this build's ``EBENLightningModule`` on the same formula weights
and clips -- the batched discriminator engine where the step takes it (both losses present), the autograd step with the redundant
passes removed, and the literal as-executed order.
"""
import os
from functools import partial

import numpy as np
import pytest
import torch

from formula import formula_state_dict, summarize
from make_variants_golden import VARIANTS, oracle_kwargs, variant_batches
from oracle import eben_oracle as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LOG_KEYS = ("train/generator/reconstructive_loss_temp", "train/generator/feature_matching_loss", "train/generator/adv_loss_gen",
            "train/generator/backprop_loss", "train/discriminator/real_loss", "train/discriminator/fake_loss",
            "train/discriminator/backprop_loss")


@pytest.fixture(scope="module")
def vgolden():
    return np.load(os.path.join(ROOT, "tests", "golden", "eben_variants_golden.npz"))


def _shapes(golden, tag):
    return {k: tuple(int(x) for x in s.split(",")) for k, s in zip(golden[f"contract/{tag}/keys"], golden[f"contract/{tag}/shapes"])}


def test_fixture_records_the_oracle_against_the_reference(vgolden):
    """make_variants_golden.py ran the oracle beside the reference in every configuration and stored the differences."""
    for name in VARIANTS:
        assert float(vgolden[f"check:var/{name}/logs_rel"]) < 5e-4, name
        assert float(vgolden[f"check:var/{name}/post_adam_maxabs"]) < 1e-3, name   # Adam's first steps: +- lr per sign flip
    # the draws of eben.py:118 as frozen: ratio 0 never updates, feature-matching-only has no discriminator phase at all
    for name, pattern in (("ratio0", [0, 0]), ("ratio05", [1, 0, 1]), ("fm_only", [0, 0]), ("none", [1, 1])):
        assert [int(vgolden[f"var/{name}/step{i}/updated"]) for i in range(len(pattern))] == pattern
    for name in ("ratio0", "fm_only"):
        assert float(vgolden[f"var/{name}/disc_moved"]) == 0.0


@pytest.mark.parametrize("name", list(VARIANTS))
def test_oracle_trainer_matches_reference_replay(golden, vgolden, name):
    cfg, steps = VARIANTS[name]
    g_shapes = _shapes(golden, "G")
    g_sd = formula_state_dict(g_shapes, "G2")
    ana, syn, _ = O.pqmf_bank(4, 32)
    g_sd["pqmf.analysis_weights"], g_sd["pqmf.synthesis_weights"] = ana, syn
    d_sd = formula_state_dict(_shapes(golden, "D"), "D")
    d0 = {k: v.clone() for k, v in d_sd.items()}
    trainer = O.OracleTrainer(g_sd, d_sd, p=2, q=4, use_mrstft=False, **oracle_kwargs(cfg))
    torch.manual_seed(int(vgolden[f"var/{name}/seed"]))
    for i, (bc, air) in enumerate(variant_batches(name, steps)):
        logs = trainer.step(bc, air)
        updated = bool(vgolden[f"var/{name}/step{i}/updated"])
        assert ("train/discriminator/backprop_loss" in logs) == updated
        for k, v in logs.items():
            if k == "enhanced":
                s = summarize(v)
                np.testing.assert_allclose(s["probe"], vgolden[f"var/{name}/step{i}/enhanced:probe"], rtol=1e-4, atol=1e-5)
            else:
                np.testing.assert_allclose(v.double().numpy(), vgolden[f"var/{name}/step{i}/{k}"], rtol=5e-4, err_msg=f"{name} step {i} {k}")
    for k, v in trainer.g.items():
        if not k.startswith("pqmf."):
            np.testing.assert_allclose(v.double().norm().item(), vgolden[f"var/{name}/post/G/{k}"][1], rtol=1e-4)
    # Adam's first steps move every element by ~lr * sign(g): where the fake and real hinge gradients cancel, the sign is decided by the
    # last bits and a small tensor's norm moves by a few 1e-4 (the reference's own fp32 / fp64 gradients differ the same way)
    for k, v in trainer.d.items():
        np.testing.assert_allclose(v.double().norm().item(), vgolden[f"var/{name}/post/D/{k}"][1], rtol=5e-4)
    if not any(bool(vgolden[f"var/{name}/step{i}/updated"]) for i in range(steps)):
        assert all(torch.equal(v.detach(), d0[k]) for k, v in trainer.d.items())   # a discriminator nobody updated is bit-identical


# ---- GPU: the product's LightningModule in the same configurations ------------------------------------------------------------------
def _make_module(golden, cfg):
    from tests.test_gpu_models import DEV, build_discriminator, build_generator
    from vibravox_amd.lightning_modules.eben import EBENLightningModule
    from vibravox_amd.optim import FusedAdam
    from vibravox_amd.torch_modules.losses.feature_loss import FeatureLossForDiscriminatorMelganMultiScales
    from vibravox_amd.torch_modules.losses.hinge_loss import HingeLossForDiscriminatorMelganMultiScales

    gen, _ = build_generator(golden, 2)
    disc, d_sd = build_discriminator(golden)
    opt = partial(FusedAdam, lr=3e-4, betas=(0.5, 0.9))
    mod = EBENLightningModule(
        sample_rate=16000, generator=gen, discriminator=disc, generator_optimizer=opt, discriminator_optimizer=opt,
        reconstructive_loss_freq_fn=None,
        reconstructive_loss_time_fn=torch.nn.L1Loss() if cfg.get("time_loss") == "l1" else None,
        feature_matching_loss_fn=FeatureLossForDiscriminatorMelganMultiScales() if cfg.get("use_fm", True) else None,
        adversarial_loss_fn=HingeLossForDiscriminatorMelganMultiScales() if cfg.get("use_adv", True) else None,
        dynamic_loss_balancing=cfg.get("balancing", "ema"), beta_ema=0.9, update_discriminator_ratio=cfg.get("ratio", 1.0))
    return mod, d_sd, DEV


@pytest.mark.gpu
@pytest.mark.parametrize("path", ["engine", "fused", "literal"])
@pytest.mark.parametrize("name", list(VARIANTS))
def test_module_step_in_the_reference_other_configurations(hip, golden, vgolden, name, path):
    """``path``: "engine" = the step as shipped (the batched discriminator engine wherever both discriminator losses are configured --
    feature-matching-only / adversarial-only take the literal order by construction), "fused" = the autograd step without the redundant
    passes, "literal" = eben.py:82-130 pass by pass.  Tolerances: those of the default-configuration replay
    (test_gpu_models.test_two_train_steps_against_reference_replay_golden)."""
    cfg, steps = VARIANTS[name]
    mod, d_sd, dev = _make_module(golden, cfg)
    mod.exploit_step_redundancy = path != "literal"
    mod.use_disc_engine = path == "engine"
    mod.disc_math = "f32"
    both = cfg.get("use_fm", True) and cfg.get("use_adv", True)
    torch.manual_seed(int(vgolden[f"var/{name}/seed"]))
    for i, (bc, air) in enumerate(variant_batches(name, steps)):
        mod.logged.clear()
        before = {k: v.clone() for k, v in mod.discriminator.state_dict().items()}
        out = mod.training_step({"audio_body_conducted": bc.to(dev), "audio_airborne": air.to(dev)})
        s = summarize(out["enhanced"].detach().cpu())
        np.testing.assert_allclose(s["probe"], vgolden[f"var/{name}/step{i}/enhanced:probe"], rtol=2e-4, atol=2e-5)
        updated = bool(vgolden[f"var/{name}/step{i}/updated"])
        for k in LOG_KEYS:
            gk = f"var/{name}/step{i}/{k}"
            assert (k in mod.logged) == (gk in vgolden.files), (name, i, k, sorted(mod.logged))
            if k in mod.logged:
                np.testing.assert_allclose(mod.logged[k].item(), vgolden[gk], rtol=2e-3 if "backprop" in k else 5e-4, err_msg=f"{name} step {i} {k}")
        if cfg.get("balancing", "ema") is not None:
            np.testing.assert_allclose(torch.stack([torch.as_tensor(t) for t in mod.last_norms]).cpu().numpy(), vgolden[f"var/{name}/step{i}/balancing/norms"], rtol=2e-3)
            np.testing.assert_allclose(torch.stack([torch.as_tensor(t) for t in mod.last_lambdas]).cpu().numpy(), vgolden[f"var/{name}/step{i}/balancing/lambdas"], rtol=2e-3)
        if not updated:   # eben.py:118 did not pass (or there is no discriminator phase): not one bit of the discriminator moves
            torch.cuda.synchronize()
            assert all(torch.equal(v, before[k]) for k, v in mod.discriminator.state_dict().items()), (name, i)
    if path == "engine" and both:
        assert getattr(mod, "_disc_engine", None) is not None, "the engine step did not run"
    for k, v in mod.generator.state_dict().items():
        if not k.startswith("pqmf."):
            np.testing.assert_allclose(v.double().norm().item(), vgolden[f"var/{name}/post/G/{k}"][1], rtol=2e-4)
    for k, v in mod.discriminator.state_dict().items():
        np.testing.assert_allclose(v.double().norm().item(), vgolden[f"var/{name}/post/D/{k}"][1], rtol=5e-4)


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["ratio05", "none", "simple"])
def test_bundle_layout_plan_in_the_other_configurations(hip, golden, vgolden, name):
    """The benchmarked plan (``bf16_bl``: bundle-layout engine, two-pass stacked backward) through the configurations its step
    special-cases: a skipped discriminator update launches neither the [fake | real] pass nor a weight gradient and leaves every
    discriminator bit where it was; no balancing / "simple" balancing change the seed arithmetic.  bf16 tolerances
    (test_gpu_models.BF16_STEP_TOLERANCES) against the reference replay."""
    from tests.test_gpu_models import BF16_STEP_TOLERANCES as TOL

    cfg, steps = VARIANTS[name]
    mod, d_sd, dev = _make_module(golden, cfg)
    mod.disc_math, mod.gen_backward_math = "bf16_bl", "bf16"
    torch.manual_seed(int(vgolden[f"var/{name}/seed"]))
    for i, (bc, air) in enumerate(variant_batches(name, steps)):
        mod.logged.clear()
        before = {k: v.clone() for k, v in mod.discriminator.state_dict().items()}
        mod.training_step({"audio_body_conducted": bc.to(dev), "audio_airborne": air.to(dev)})
        torch.cuda.synchronize()
        updated = bool(vgolden[f"var/{name}/step{i}/updated"])
        assert ("train/discriminator/backprop_loss" in mod.logged) == updated
        moved = max(float((v - before[k]).abs().max()) for k, v in mod.discriminator.state_dict().items())
        assert (moved > 0) == updated, (name, i, moved)
        for k in LOG_KEYS:
            if k in mod.logged:
                rtol = TOL["feature_matching_loss"] if "feature_matching" in k else TOL["backprop_loss"] if "generator/backprop" in k else TOL["loss"]
                # after a discriminator update the bf16 and fp32 trajectories have parted by one Adam step: the later steps get the wide bar
                np.testing.assert_allclose(mod.logged[k].item(), vgolden[f"var/{name}/step{i}/{k}"], rtol=rtol if i == 0 else 5e-2, err_msg=f"{name} step {i} {k}")
    assert type(mod._disc_engine).__name__ == "DiscriminatorEngineBL"


@pytest.mark.gpu
def test_grouped_weight_gradients_equal_chain_by_chain(hip, golden):
    """The three PQMF-band chains' weight gradients as one launch sequence (EBEN_DW_GROUP, ``_ChainBL.weight_grads_group``) and the
    two-pass stacked backward (EBEN_SPLIT_BWD) against chain by chain / one 4B-row pass: the discriminator after one step is bit-identical
    (same kernels on the same operands; only the launches they travel in differ)."""
    from tests.test_gpu_models import DEV
    from vibravox_amd.disc_engine_bl import DiscriminatorEngineBL

    outs = []
    saved = DiscriminatorEngineBL.group_weight_grads, DiscriminatorEngineBL.split_backward
    try:
        for group, split in ((True, True), (False, True), (False, False)):
            DiscriminatorEngineBL.group_weight_grads, DiscriminatorEngineBL.split_backward = group, split
            mod, _, dev = _make_module(golden, {})
            mod.disc_math, mod.gen_backward_math = "bf16_bl", "bf16"
            torch.manual_seed(3)
            for i, (bc, air) in enumerate(variant_batches("none", 2)):
                mod.training_step({"audio_body_conducted": bc.to(dev), "audio_airborne": air.to(dev)})
            torch.cuda.synchronize()
            outs.append({k: v.clone() for k, v in mod.discriminator.state_dict().items()})
    finally:
        DiscriminatorEngineBL.group_weight_grads, DiscriminatorEngineBL.split_backward = saved
    for other in outs[1:]:
        assert all(torch.equal(outs[0][k], other[k]) for k in outs[0]), [k for k in outs[0] if not torch.equal(outs[0][k], other[k])]
