"""EBEN GAN training orchestration on the HIP modules.

Drop-in for ``vibravox/lightning_modules/eben.py:9-240`` (``EBENLightningModule``): same constructor
arguments and asserts (:10-80), the same phase order in ``training_step`` (:82-130) -- generator
forward, atomic losses logged *before* balancing, dynamic loss balancing on the gradient norms at
``generator.last_conv.weight`` incl. the first-call EMA quirk (:222-240), manual backward + Adam,
then the discriminator phase on detached generator outputs with the ``torch.rand(1) < ratio`` draw
(:118) -- the same logged names and the same returned dict.

``exploit_step_redundancy`` (default on) removes work the reference repeats without changing any
value:
  * the two discriminator forwards of the discriminator phase recompute, with unchanged
    discriminator weights, exactly what the generator phase computed on the same tensors -> the
    hinge losses are taken from the generator-phase embeddings and back-propagated with
    ``backward(inputs=discriminator parameters)``;
  * the generator only reaches the losses through ``bands`` (``enhanced`` = PQMF synthesis of it), so
    the three balancing gradients are taken at ``bands`` (one discriminator input-gradient pass each
    for feature-matching and adversarial, none repeated), pushed to ``last_conv.weight`` for the
    norms, and the final generator backward is seeded with their lambda-weighted sum (linearity)
    instead of walking the discriminator a third time.
11 discriminator passes per step become 8 (F_exec -> F_min of SURVEY.md section 8d).  Set the
attribute to False for the literal as-executed order.
"""
from __future__ import annotations

import inspect
import os
from functools import partial
from typing import Dict, Optional

import torch

from .. import ops
from .base_se import BaseSELightningModule


def _takes_grad_scale(optimizer) -> bool:
    """Whether ``optimizer.step`` accepts ``grad_scale`` -- decided from the signature of the innermost optimiser, never by
    catching a TypeError raised somewhere inside a step that may already have updated parameters."""
    inner = optimizer
    while hasattr(inner, "optimizer") and not isinstance(inner, torch.optim.Optimizer):
        inner = inner.optimizer   # ddp.BucketedZeroGrad / Lightning's optimizer wrapper
    try:
        return "grad_scale" in inspect.signature(type(inner).step).parameters
    except (TypeError, ValueError):
        return False


#: ``disc_math`` names -> what ``DiscriminatorEngine`` runs (see its ``math`` argument).
DISC_MATH_PLANS = {
    "f32": ops.MATH_F32,
    "bf16_plain": ops.MATH_BF16,
    # BASELINE config 2: bf16 MFMA operands everywhere except the FORWARD of the three PQMF-band discriminators, whose operands
    # enter as hi + lo bf16 pieces (three piece products, ~2^-17: EBEN_MATH_BF16X3).  [MI355X] discriminator gradient against the
    # fp32 step at config 2: 3.378e-2 with that forward in exact fp32, in six-piece (fp32-grade) or in three-piece products alike --
    # what is left comes from MelGAN's bf16 forward and the bf16 gradient contractions; 0.18 with every contraction on single bf16.
    "bf16": {"pqmf": (ops.MATH_BF16X3, ops.MATH_BF16, ops.MATH_BF16), "melgan": ops.MATH_BF16},
    # the same plan with every embedding / stacked gradient of the engine at rest in the bf16 bundle layout (disc_engine_bl.py): the
    # MFMA operands are the same roundings of the same fp32 values, produced by the writer's epilogue instead of each reader's staging
    "bf16_bl": {"pqmf": (ops.MATH_BF16X3, ops.MATH_BF16, ops.MATH_BF16), "melgan": ops.MATH_BF16, "layout": "bl"},
    "bf16_f32fwd": {"pqmf": (ops.MATH_F32, ops.MATH_BF16, ops.MATH_BF16), "melgan": ops.MATH_BF16},
    "bf16x2": (ops.MATH_BF16X2, ops.MATH_BF16, ops.MATH_BF16X2),
    # fp32 arithmetic on the bf16 matrix pipe: forward and input gradients with both operands as three bf16 pieces (six piece
    # products, dropped terms <= 2^-26: tapconv3.hip), weight gradients on the exact-fp32 kernels
    "bf16x6": (ops.MATH_BF16X6, ops.MATH_BF16X6, ops.MATH_F32),
    # the "bf16" plan with the PQMF-band forwards in that fp32-grade split form instead of the fp32 MFMA
    "bf16_x6fwd": {"pqmf": (ops.MATH_BF16X6, ops.MATH_BF16, ops.MATH_BF16), "melgan": ops.MATH_BF16},
}

#: generator-side loss terms in the reference's insertion order (eben.py:195-211): logged name, the module attribute that holds
#: the loss function, and what it is evaluated on
_GENERATOR_TERMS = (
    ("reconstructive_loss_freq", "reconstructive_loss_freq_fn", "waveforms"),
    ("reconstructive_loss_temp", "reconstructive_loss_temp_fn", "waveforms"),
    ("feature_matching_loss", "feature_matching_loss_fn", "both embeddings"),
    ("adv_loss_gen", "adversarial_loss_fn", "enhanced embeddings"),
)
#: discriminator-side terms (eben.py:212-219): logged name, branch, hinge target
_DISCRIMINATOR_TERMS = (("real_loss", "reference", 1), ("fake_loss", "enhanced", -1))


class EBENLightningModule(BaseSELightningModule):
    exploit_step_redundancy: bool = True

    def __init__(
        self,
        sample_rate: int,
        generator: torch.nn.Module,
        discriminator: torch.nn.Module,
        generator_optimizer: "partial[torch.optim.Optimizer]",
        discriminator_optimizer: "partial[torch.optim.Optimizer]",
        reconstructive_loss_freq_fn: torch.nn.Module = None,
        reconstructive_loss_time_fn: torch.nn.Module = None,
        feature_matching_loss_fn: torch.nn.Module = None,
        adversarial_loss_fn: torch.nn.Module = None,
        dynamic_loss_balancing: str = None,
        beta_ema: float = 0.9,
        update_discriminator_ratio: float = 1.0,
        description: str = None,
        push_to_hub_after_testing: bool = False,
    ):
        super().__init__(sample_rate=sample_rate, description=description)
        self.generator = generator
        self.discriminator = discriminator
        self.generator_optimizer = generator_optimizer(params=self.generator.parameters())
        self.discriminator_optimizer = discriminator_optimizer(params=self.discriminator.parameters())
        self.reconstructive_loss_temp_fn = reconstructive_loss_time_fn
        self.reconstructive_loss_freq_fn = reconstructive_loss_freq_fn
        self.feature_matching_loss_fn = feature_matching_loss_fn
        self.adversarial_loss_fn = adversarial_loss_fn
        assert dynamic_loss_balancing in {None, "simple", "ema"}, "dynamic_loss_balancing must be in {None, 'simple', 'ema'}"
        self.dynamic_loss_balancing = dynamic_loss_balancing
        self.atomic_norms_old = None
        self.beta_ema = beta_ema
        assert 0 <= update_discriminator_ratio <= 1, "update_discriminator_ratio must be in [0, 1]"
        self.update_discriminator_ratio = update_discriminator_ratio
        self.push_to_hub_after_testing = push_to_hub_after_testing
        self.automatic_optimization = False
        self.last_lambdas = None
        self.last_norms = None

    def configure_optimizers(self):
        return [self.generator_optimizer, self.discriminator_optimizer]

    # -- data-parallel hook points (no-ops on one GPU) --------------------------------------
    def _sync_grads(self, optimizer) -> float:
        sync = getattr(self, "grad_sync", {}).get(id(optimizer)) if hasattr(self, "grad_sync") else None
        return sync.finish() if sync is not None else 1.0

    def _step(self, optimizer, grad_scale: float):
        """optimizer.step() on gradient SUMS over ranks: ``grad_scale`` (1 / world size) is folded into the Adam kernel where
        the optimiser takes it (``FusedAdam.step(grad_scale=)``, also behind ``BucketedZeroGrad``), else applied in place."""
        if grad_scale != 1.0:
            if _takes_grad_scale(optimizer):
                optimizer.step(grad_scale=grad_scale)
                return
            for grp in optimizer.param_groups:
                for p in grp["params"]:
                    if p.grad is not None:
                        p.grad.mul_(grad_scale)
        optimizer.step()

    def training_step(self, batch: Dict[str, torch.Tensor], batch_idx: int = 0):
        if self.exploit_step_redundancy and self.adversarial_loss_fn is not None and self.feature_matching_loss_fn is not None:
            step = self._training_step_engine if self._engine_usable(batch) else self._training_step_fused
        else:
            step = self._training_step_literal
        out = step(batch)
        flush = getattr(self, "flush_logged", None)   # Lightning-free stand-in: the step's sync_dist values as one collective
        if flush is not None:
            flush()
        ops.capture_gate.step_end()   # multi-rank: the ranks' vote on graph captures in the next step (a no-op on one rank / once settled)
        return out

    #: run the discriminator passes batched and outside autograd (vibravox_amd/disc_engine.py)
    use_disc_engine: bool = os.environ.get("EBEN_DISC_ENGINE", "1") != "0"
    #: arithmetic of the discriminator contractions inside the engine: a key of DISC_MATH_PLANS -- "f32" (bit-exact fp32
    #: products), "bf16" (bf16 MFMA operands, fp32 accumulate, activation operand split -- BASELINE config 2), "bf16_plain" --
    #: or a plan the engine understands.  The generator's forward computes in fp32-grade split-bf16 products (gen_engine.RU_FWD_MATH /
    #: CONV_FWD_MATH = EBEN_MATH_BF16X6: every mantissa bit of both fp32 operands, fp32 accumulation) -- except INSIDE the engine
    #: train step of the bf16-mixed plan (gen_backward_math "bf16" and ``ru_forward_x3``), whose forward takes hi + lo operands
    #: (EBEN_MATH_BF16X3, ~2^-17 per product); validation / prediction forwards are fp32-grade in every plan.
    disc_math: str = os.environ.get("EBEN_DISC_MATH", "f32")
    #: arithmetic of the generator's BACKWARD contractions (input / weight gradients) in the engine step; "bf16" also moves the step's
    #: forward to hi + lo operands (see ``ru_forward_x3``; "f32": six-piece bf16 products, <= 2^-26 dropped per product: within 2x of
    #: the fp32 MFMA kernels' own error against fp64, tests/test_gpu_ops.py; EBEN_RU_FWD_MATH=f32 selects those kernels)
    gen_backward_math: str = os.environ.get("EBEN_GEN_BWD_MATH", "f32")

    #: arithmetic of the MRSTFT loss's windowed-DFT contractions in the engine step (mrstft_loss.MultiResolutionSTFTLoss.stft_math):
    #: None leaves the loss module's own setting (exact fp32, folded); "bf16x3" goes with the bf16 step of BASELINE config 2
    stft_math: Optional[str] = os.environ.get("EBEN_STEP_STFT_MATH") or None

    #: ``trainer.precision`` (vibravox configs/trainer/ddp.yaml:23-25, Lightning's names) -> (disc_math, gen_backward_math, stft_math).
    #: "32-true" is the reference's arithmetic (exact fp32 products), "bf16-mixed" the plan BASELINE config 2 names and bench.py
    #: measures (bf16 MFMA operands, fp32 accumulate / parameters / optimiser state), "32-split" fp32-grade arithmetic on the bf16
    #: matrix pipe (six bf16 piece products per fp32 product).
    PRECISION_PLANS = {
        "32-true": ("f32", "f32", None),
        "bf16-mixed": ("bf16_bl", "bf16", "folded_x3"),
        "32-split": ("bf16x6", "f32", "folded_x6"),
    }
    _PRECISION_ALIASES = {"32": "32-true", "fp32": "32-true", "f32": "32-true", "bf16": "bf16-mixed", "bf16-true": "bf16-mixed",
                          "bf16x6": "32-split"}

    def set_precision(self, precision) -> "EBENLightningModule":
        """Selects the step's arithmetic plan from the trainer's ``precision`` key (``run.py`` passes ``trainer.precision`` here)."""
        key = str(precision).strip().lower()
        if key in ("16-mixed", "16", "fp16", "16-true"):
            # Lightning's fp16 AMP: no such plan here (the MFMA operands are bf16; mapping it silently would misreport the arithmetic)
            raise ValueError(f"precision {precision!r} (fp16) is not offered: use 'bf16-mixed', '32-split' or '32-true'")
        key = self._PRECISION_ALIASES.get(key, key)
        if key not in self.PRECISION_PLANS:
            raise ValueError(f"precision must be one of {sorted(self.PRECISION_PLANS)} (or {sorted(self._PRECISION_ALIASES)}), got {precision!r}")
        overridden = [v for v in ("EBEN_DISC_MATH", "EBEN_GEN_BWD_MATH", "EBEN_STEP_STFT_MATH") if os.environ.get(v)]
        if overridden:
            import warnings
            warnings.warn(f"trainer.precision={key!r} overrides {', '.join(overridden)} (set by the environment)", stacklevel=2)
        self.disc_math, self.gen_backward_math, stft = self.PRECISION_PLANS[key]
        # "32-true": the loss module's OWN arithmetic again (the step writes stft_math into the module when it is not None: a module that
        # ran a bf16-mixed step must not keep folded_x3 while the step reports the reference's exact fp32)
        loss = getattr(self, "reconstructive_loss_freq_fn", None)
        if loss is not None and hasattr(loss, "stft_math"):
            if not hasattr(self, "_loss_stft_math0"):
                self._loss_stft_math0 = loss.stft_math
            if stft is None:
                loss.stft_math = self._loss_stft_math0
        self.stft_math = stft
        self.precision = key
        return self

    def _engine_usable(self, batch) -> bool:
        from ..disc_engine import DiscriminatorEngine
        from ..torch_modules.losses.feature_loss import FeatureLossForDiscriminatorMelganMultiScales
        from ..torch_modules.losses.hinge_loss import HingeLossForDiscriminatorMelganMultiScales

        return (self.use_disc_engine and batch["audio_airborne"].is_cuda and DiscriminatorEngine.supports(self.discriminator)
                and type(self.feature_matching_loss_fn) is FeatureLossForDiscriminatorMelganMultiScales
                and type(self.adversarial_loss_fn) is HingeLossForDiscriminatorMelganMultiScales)

    #: tools/phase_times.py: list that receives (label, event) pairs recorded on the main stream between the phases
    phase_events = None

    #: tools/host_times.py: list that receives (label, time.perf_counter()) at the same points -- where the HOST spends the step
    phase_host = None

    def _mark(self, label: str) -> None:
        if self.phase_host is not None:
            import time
            self.phase_host.append((label, time.perf_counter()))
        if self.phase_events is not None:
            ev = torch.cuda.Event(enable_timing=True)
            ev.record()
            self.phase_events.append((label, ev))

    def _training_step_engine(self, batch: Dict[str, torch.Tensor]):
        # bf16-mixed: the generator forward on hi + lo operands (three piece products, 2^-17 each) instead of the fp32-grade six -- the
        # output stays orders of magnitude inside north_star's 1e-5 MSE (tests: < 1e-8 against the oracle at config 2).  Scoped to the
        # step (its forward and the weight-image rebuilds behind the optimiser steps): evaluation forwards keep the default arithmetic.
        from .. import gen_engine
        with gen_engine.forward_math("bf16x3" if self.gen_backward_math == "bf16" and self.ru_forward_x3 else None):
            return self._training_step_engine_body(batch)

    def _training_step_engine_body(self, batch: Dict[str, torch.Tensor]):
        """``_training_step_fused`` with the discriminator side run by ``DiscriminatorEngine``: one
        batch-2B forward for the enhanced and reference branches and one stacked backward for the four
        gradient signals (feature matching, adversarial, fake, real).  Same logged values, same
        parameter updates (the weight-gradient sums are taken in a different fp32 order)."""
        from ..disc_engine import DiscriminatorEngine, inject_grads

        corrupted_speech = self.generator.cut_to_valid_length(batch["audio_body_conducted"])
        reference_speech = self.generator.cut_to_valid_length(batch["audio_airborne"])
        generator_optimizer, discriminator_optimizer = self.optimizers(use_pl_optimizer=True)
        g_params = [p for p in ops.parameters_of(self.generator) if p.requires_grad]
        math = DISC_MATH_PLANS[self.disc_math] if isinstance(self.disc_math, str) else self.disc_math
        if getattr(self, "_disc_engine", None) is None or self._disc_engine.disc is not self.discriminator or self._disc_engine.math != math:
            self._disc_engine = DiscriminatorEngine(self.discriminator, math)
        engine = self._disc_engine

        if self.stft_math is not None and hasattr(self.reconstructive_loss_freq_fn, "stft_math"):
            self.reconstructive_loss_freq_fn.stft_math = self.stft_math

        # ---- generator phase
        ops.join_prepack()
        self._mark("start")
        with torch.no_grad():
            bands_ref = self.generator.pqmf.forward(reference_speech, "analysis")
        if self.split_discriminator_forward and (type(engine).__name__ == "DiscriminatorEngineBL" or self._split_forward_forced):
            # the reference half of the discriminator batch does not depend on the generator: it runs on the chains'
            # streams underneath the generator forward (a chain of small launches that leaves most of the GPU idle).
            # Measured for the bundle-layout engine only (0.2 ms of a 10 ms step); the fp32-at-rest engines keep the
            # one-piece forward unless EBEN_SPLIT_D_FWD=1 asks for it.
            engine.forward_reference(bands_ref, reference_speech)
        with ops.backward_math({"f32": ops.MATH_F32, "bf16": ops.MATH_BF16}[self.gen_backward_math]):
            enhanced_speech, bands = self.generator(corrupted_speech)
        self._mark("generator forward")
        # the four discriminator chains start on their streams; the reconstructive losses (which do not involve the
        # discriminators) run on this stream underneath them
        engine.forward(bands.detach(), enhanced_speech.detach(), bands_ref, reference_speech, join=False)
        losses: Dict[str, torch.Tensor] = {}
        if self.reconstructive_loss_freq_fn:
            losses["reconstructive_loss_freq"] = self.reconstructive_loss_freq_fn(enhanced_speech, reference_speech)
        if self.reconstructive_loss_temp_fn:
            losses["reconstructive_loss_temp"] = self.reconstructive_loss_temp_fn(enhanced_speech, reference_speech)
        self._mark("reconstructive losses (discriminator forward beside them)")
        engine.join()
        self._mark("discriminator forward joined")
        d_losses = engine.losses()
        self._mark("fm + hinge losses")
        losses["feature_matching_loss"] = d_losses["feature_matching_loss"]
        losses["adv_loss_gen"] = d_losses["adv_loss_gen"]
        for key, value in losses.items():
            self.log(f"train/generator/{key}", value, sync_dist=True)

        # the discriminator-phase draw (eben.py:118) is the only CPU RNG use of the step: taking it here
        # leaves the sequence of draws unchanged and lets the stacked backward skip the weight gradients
        update_discriminator = bool(torch.rand(1) < self.update_discriminator_ratio)
        # input gradients now; the discriminator's weight gradients keep running on the engine's streams
        # underneath the balancing passes and the (launch-bound, GPU-underfilling) generator backward
        syncs = getattr(self, "grad_sync", None) or {}
        g_sink, d_sink = syncs.get(id(generator_optimizer)), syncs.get(id(discriminator_optimizer))
        engine.backward_launch(want_param_grads=update_discriminator, sink=d_sink)
        # balancing (eben.py:222-240) with every gradient taken at `bands`; the seeds of the losses that do not pass
        # through the discriminators are taken while its input-gradient chains run
        leaf = self.generator.last_conv.weight
        own = {key: torch.autograd.grad(loss, bands, retain_graph=True)[0] for key, loss in losses.items()
               if key not in ("feature_matching_loss", "adv_loss_gen")}
        self._mark("reconstructive seeds (discriminator input gradients beside them)")
        fm_b, fm_a, adv_b, adv_a = engine.backward_finish()
        self._mark("discriminator input gradients joined")
        seeds = []
        for key, loss in losses.items():
            if key == "feature_matching_loss":
                seeds.append(fm_b + torch.autograd.grad(enhanced_speech, bands, grad_outputs=fm_a, retain_graph=True)[0])
            elif key == "adv_loss_gen":
                seeds.append(adv_b + torch.autograd.grad(enhanced_speech, bands, grad_outputs=adv_a, retain_graph=True)[0])
            else:
                seeds.append(own[key])
        atomic_norms = None
        pre = getattr(self.generator, "_last_pre", None)
        if self.dynamic_loss_balancing is None:
            pass   # eben.py:107-108: no balancing -- no norms, every lambda is 1
        elif self.fused_norms and pre is not None and pre.shape[0] == bands.shape[0] and pre.shape[2] == bands.shape[2]:
            atomic_norms = ops.last_conv_grad_norms(seeds, bands, pre, self.generator.last_conv)   # one pass for the three losses
        if atomic_norms is None and self.dynamic_loss_balancing is not None:
            with ops.input_grads_disabled():   # only last_conv's weight gradient is wanted: its input gradient would be computed and dropped
                atomic_norms = [torch.norm(torch.autograd.grad(bands, leaf, grad_outputs=s, retain_graph=True)[0]).detach() for s in seeds]
        if self.dynamic_loss_balancing is None:
            backprop_loss_generator = sum(loss.detach() for loss in losses.values())
            seed = seeds[0] if len(seeds) == 1 else ops.weighted_sum(seeds, self._unit_lambdas(len(seeds), seeds[0].device))
        elif self.fused_balancing and len(seeds) <= 8 and atomic_norms[0].is_cuda:
            lambdas, backprop_loss_generator = self._update_lambdas_fused(atomic_norms, [loss.detach() for loss in losses.values()])
            seed = ops.weighted_sum(seeds, self._bal["lam"])
        else:
            lambdas = self._update_lambdas(atomic_norms)
            backprop_loss_generator = sum(loss.detach() * lam for loss, lam in zip(losses.values(), lambdas))
            seed = None
            for s, lam in zip(seeds, lambdas):
                seed = s * lam if seed is None else seed + s * lam
        self.log("train/generator/backprop_loss", backprop_loss_generator, sync_dist=True)
        self._mark("balancing (3 seeds + norms)")
        with ops.weight_grads_on_side_stream(sink=g_sink) as side:   # dX chain on this stream, dW work beside it
            torch.autograd.backward(bands, seed, inputs=g_params)
        self._mark("generator backward (dX chain)")
        side.join()
        self._mark("generator weight gradients joined")
        self._step(generator_optimizer, self._sync_grads(generator_optimizer))
        generator_optimizer.zero_grad()
        self._mark("generator Adam")
        if self.prepack_weights:
            ops.prepack(self._hip_convs(self.generator))   # next step's generator images, under the discriminator phase
            if getattr(self.generator, "_engine", None) is not None:
                self.generator._engine.prepack()

        # ---- discriminator phase: the gradients of real_loss + fake_loss are already there
        if update_discriminator:
            real_loss, fake_loss = d_losses["real_loss"], d_losses["fake_loss"]
            self.log("train/discriminator/real_loss", real_loss, sync_dist=True)
            self.log("train/discriminator/fake_loss", fake_loss, sync_dist=True)
            self.log("train/discriminator/backprop_loss", real_loss + fake_loss, sync_dist=True)
            d_grads = engine.collect_param_grads()   # None when they went straight into the data-parallel buckets
            if d_grads is not None:
                inject_grads(ops.parameters_of(self.discriminator), d_grads)
            self._mark("discriminator weight gradients joined")
            self._step(discriminator_optimizer, self._sync_grads(discriminator_optimizer))
            discriminator_optimizer.zero_grad()
            self._mark("discriminator Adam")
            if self.prepack_weights:
                engine.prepack()   # next step's discriminator images, under the next generator forward
        return {"corrupted": corrupted_speech, "enhanced": enhanced_speech.detach(), "reference": reference_speech}

    #: run the discriminators on the reference half of the batch underneath the generator forward (it does not depend on the generator).
    #: Round 1 (fp32 discriminators): neutral, 25.8 vs 25.8 ms/step.  Round 4 (bundle-layout plan, generator forward a chain of latency-bound
    #: launches): [MI355X, same box, interleaved] 10.25 -> 10.05 ms/step -- the generator forward phase lengthens by 0.7 ms (1.47 -> 2.19: its
    #: launches share the CUs), the discriminator forward phase shortens by 0.8 (2.26 -> 1.49).  On by default.
    split_discriminator_forward: bool = os.environ.get("EBEN_SPLIT_D_FWD", "1") != "0"
    _split_forward_forced: bool = os.environ.get("EBEN_SPLIT_D_FWD") == "1"

    #: bf16-mixed plan: ResidualUnit forwards with three piece products (EBEN_MATH_BF16X3) instead of six; EBEN_RU_FWD_X3=0 keeps six
    ru_forward_x3: bool = (os.environ.get("EBEN_RU_FWD_X3", "1") != "0" and not os.environ.get("EBEN_RU_FWD_MATH")
                           and not os.environ.get("EBEN_GEN_CONV_FWD_MATH"))

    #: rebuild the packed weight images right after each optimiser step, on the side stream (off the critical path)
    prepack_weights: bool = os.environ.get("EBEN_PREPACK", "1") != "0"

    def _hip_convs(self, net):
        cached = getattr(self, "_hip_conv_cache", None)
        if cached is None or cached[0] is not net:
            from ..torch_modules.utils import HipConv1d
            cached = self._hip_conv_cache = (net, [m for m in net.modules() if isinstance(m, HipConv1d)])
        return cached[1]

    def _training_step_fused(self, batch: Dict[str, torch.Tensor]):
        """Same values as ``_training_step_literal`` with 8 instead of 11 discriminator passes."""
        corrupted_speech = self.generator.cut_to_valid_length(batch["audio_body_conducted"])
        reference_speech = self.generator.cut_to_valid_length(batch["audio_airborne"])
        generator_optimizer, discriminator_optimizer = self.optimizers(use_pl_optimizer=True)
        g_params = [p for p in ops.parameters_of(self.generator) if p.requires_grad]
        d_params = [p for p in ops.parameters_of(self.discriminator) if p.requires_grad]

        # ---- generator phase: every forward of the step happens here
        enhanced_speech, bands = self.generator(corrupted_speech)
        bands_ref = self.generator.pqmf.forward(reference_speech, "analysis")
        losses: Dict[str, torch.Tensor] = {}
        if self.reconstructive_loss_freq_fn:
            losses["reconstructive_loss_freq"] = self.reconstructive_loss_freq_fn(enhanced_speech, reference_speech)
        if self.reconstructive_loss_temp_fn:
            losses["reconstructive_loss_temp"] = self.reconstructive_loss_temp_fn(enhanced_speech, reference_speech)
        enhanced_embeddings = self.discriminator(bands=bands, audio=enhanced_speech)
        reference_embeddings = self.discriminator(bands=bands_ref, audio=reference_speech)
        losses["feature_matching_loss"] = self.feature_matching_loss_fn(enhanced_embeddings, reference_embeddings)
        losses["adv_loss_gen"] = self.adversarial_loss_fn(embeddings=enhanced_embeddings, target=1)
        for key, value in losses.items():
            self.log(f"train/generator/{key}", value, sync_dist=True)

        # balancing (eben.py:222-240) with the gradients taken at `bands`
        leaf = self.generator.last_conv.weight
        with ops.weight_grads_disabled():  # only d/d(bands) is wanted from these passes
            seeds = [torch.autograd.grad(loss, bands, retain_graph=True)[0] for loss in losses.values()]
        if self.dynamic_loss_balancing is None:   # eben.py:107-108: the plain sum
            lambdas = [1.0] * len(seeds)
        else:
            atomic_norms = [torch.norm(torch.autograd.grad(bands, leaf, grad_outputs=s, retain_graph=True)[0]).detach() for s in seeds]
            lambdas = self._update_lambdas(atomic_norms)
        backprop_loss_generator = sum(loss.detach() * lam for loss, lam in zip(losses.values(), lambdas))
        self.log("train/generator/backprop_loss", backprop_loss_generator, sync_dist=True)
        seed = None
        for s, lam in zip(seeds, lambdas):
            seed = s * lam if seed is None else seed + s * lam
        torch.autograd.backward(bands, seed, inputs=g_params, retain_graph=True)
        self._step(generator_optimizer, self._sync_grads(generator_optimizer))
        generator_optimizer.zero_grad()

        # ---- discriminator phase on the embeddings already computed (weights untouched since)
        if torch.rand(1) < self.update_discriminator_ratio:
            real_loss = self.adversarial_loss_fn(embeddings=reference_embeddings, target=1)
            fake_loss = self.adversarial_loss_fn(embeddings=enhanced_embeddings, target=-1)
            self.log("train/discriminator/real_loss", real_loss, sync_dist=True)
            self.log("train/discriminator/fake_loss", fake_loss, sync_dist=True)
            backprop_loss_discriminator = real_loss + fake_loss
            self.log("train/discriminator/backprop_loss", backprop_loss_discriminator, sync_dist=True)
            backprop_loss_discriminator.backward(inputs=d_params)
            self._step(discriminator_optimizer, self._sync_grads(discriminator_optimizer))
            discriminator_optimizer.zero_grad()
        return {"corrupted": corrupted_speech, "enhanced": enhanced_speech.detach(), "reference": reference_speech}

    def _training_step_literal(self, batch: Dict[str, torch.Tensor]):
        """The reference's as-executed order (eben.py:82-130): each network's phase evaluates its own atomic losses from
        scratch -- 4 discriminator forwards, 3 balancing backward passes + the final one."""
        generator_optimizer, discriminator_optimizer = self.optimizers(use_pl_optimizer=True)
        corrupted = self.generator.cut_to_valid_length(batch["audio_body_conducted"])
        reference = self.generator.cut_to_valid_length(batch["audio_airborne"])

        def phase(network, optimizer, signals, gate):
            # one network's turn: freeze the other one, losses -> log -> (balance) -> backward -> Adam -> unfreeze.  `gate`
            # decides, AFTER the forwards and BEFORE anything is logged, whether this phase updates at all (eben.py:118).
            self.toggle_optimizer(optimizer)
            terms = self.compute_atomic_losses(network, *signals)
            if gate(terms):
                for name, value in terms.items():
                    self.log(f"train/{network}/{name}", value, sync_dist=True)
                if network == "generator" and self.dynamic_loss_balancing is not None:
                    terms = self.dynamically_balance_losses(terms)
                total = sum(terms.values())
                self.log(f"train/{network}/backprop_loss", total, sync_dist=True)
                self.manual_backward(total)
                self._step(optimizer, self._sync_grads(optimizer))
                optimizer.zero_grad()
            self.untoggle_optimizer(optimizer)

        enhanced, bands = self.generator(corrupted)   # no discriminator parameter involved: same graph inside or outside the toggle
        signals = (enhanced, reference, bands, self.generator.pqmf.forward(reference, "analysis"))
        phase("generator", generator_optimizer, signals, lambda terms: True)
        phase("discriminator", discriminator_optimizer, signals,
              lambda terms: bool(terms) and bool(torch.rand(1) < self.update_discriminator_ratio))
        return {"corrupted": corrupted, "enhanced": enhanced, "reference": reference}

    # -- evaluation -----------------------------------------------------
    def common_eval_step(self, batch: Dict[str, torch.Tensor], batch_idx: int, stage: str, dataloader_idx: int = 0):
        """eben.py:132-165: generator forward on the cut clip; with a reference in the batch also the atomic
        losses of both networks, logged as ``{stage}/{network}/{loss}[/{dataloader}]``.  (The reference's
        metric / audio logging of ``base_se.py:67-130`` is torchmetrics / torchaudio code outside this path.)"""
        with torch.no_grad():
            corrupted_speech = self.generator.cut_to_valid_length(batch["audio_body_conducted"])
            enhanced_speech, decomposed_enhanced_speech = self.generator(corrupted_speech)
            outputs = {"corrupted": corrupted_speech, "enhanced": enhanced_speech}
            if "audio_airborne" in batch:
                reference_speech = self.generator.cut_to_valid_length(batch["audio_airborne"])
                decomposed_reference_speech = self.generator.pqmf.forward(reference_speech, "analysis")
                outputs["reference"] = reference_speech
                names = getattr(self, "dataloader_names", None)
                dl_name = f"/{names[dataloader_idx]}" if names else ""
                for net_type in ["generator", "discriminator"]:
                    atomic_losses = self.compute_atomic_losses(net_type, enhanced_speech, reference_speech,
                                                               decomposed_enhanced_speech, decomposed_reference_speech)
                    for key, value in atomic_losses.items():
                        self.log(f"{stage}/{net_type}/{key}{dl_name}", value, sync_dist=True, add_dataloader_idx=False)
        return outputs

    def validation_step(self, batch, batch_idx: int = 0, dataloader_idx: int = 0):
        return self.common_eval_step(batch, batch_idx, "validation", dataloader_idx)

    def test_step(self, batch, batch_idx: int = 0, dataloader_idx: int = 0):
        return self.common_eval_step(batch, batch_idx, "test", dataloader_idx)

    def compute_atomic_losses(self, network, enhanced_speech, reference_speech, decomposed_enhanced_speech,
                              decomposed_reference_speech) -> Dict[str, torch.Tensor]:
        """eben.py:184-220 as two term tables: the generator's losses on the live graph, the discriminator's hinge terms on
        detached generator outputs.  Each discriminator forward is run at most once per call and only if a term needs it."""
        assert network in {"generator", "discriminator"}
        detached = network == "discriminator"
        branch_inputs = {
            "enhanced": (decomposed_enhanced_speech.detach() if detached else decomposed_enhanced_speech,
                         enhanced_speech.detach() if detached else enhanced_speech),
            "reference": (decomposed_reference_speech, reference_speech),
        }
        cache = {}

        def embeddings(branch):
            if branch not in cache:
                bands, audio = branch_inputs[branch]
                cache[branch] = self.discriminator(bands=bands, audio=audio)
            return cache[branch]

        out: Dict[str, torch.Tensor] = {}
        if network == "generator":
            if self.feature_matching_loss_fn or self.adversarial_loss_fn:
                embeddings("enhanced")   # the reference runs the enhanced branch first (eben.py:204)
            for name, attr, operands in _GENERATOR_TERMS:
                fn = getattr(self, attr)
                if not fn:
                    continue
                if operands == "waveforms":
                    out[name] = fn(enhanced_speech, reference_speech)
                elif operands == "both embeddings":
                    out[name] = fn(embeddings("enhanced"), embeddings("reference"))
                else:
                    out[name] = fn(embeddings=embeddings("enhanced"), target=1)
        elif self.adversarial_loss_fn:
            embeddings("enhanced")       # eben.py:214 before :217
            for name, branch, target in _DISCRIMINATOR_TERMS:
                out[name] = self.adversarial_loss_fn(embeddings=embeddings(branch), target=target)
        return out

    def _unit_lambdas(self, n: int, device) -> torch.Tensor:
        """n ones on the device (the weighted seed sum of a step without balancing)."""
        ones = getattr(self, "_ones", None)
        if ones is None or ones.numel() != n or ones.device != device:
            ones = self._ones = torch.ones(n, dtype=torch.float32, device=device)
        return ones

    def _update_lambdas(self, atomic_norms):
        """Loss weights from the gradient norms at ``generator.last_conv.weight`` (eben.py:229-237): the state is initialised
        with the first norms and -- quirk kept -- the EMA update is applied on that same call; rank-local like the reference's."""
        if self.atomic_norms_old is None or self.dynamic_loss_balancing == "simple":
            self.atomic_norms_old = atomic_norms
        if self.dynamic_loss_balancing == "ema":
            self.atomic_norms_old = [self.beta_ema * old + (1 - self.beta_ema) * new for old, new in zip(self.atomic_norms_old, atomic_norms)]
        lambdas = [torch.clamp(1 / (norm + 1e-4), min=0.0, max=1e4) for norm in self.atomic_norms_old]
        self.last_norms, self.last_lambdas = atomic_norms, lambdas
        return lambdas

    #: the engine step's balancing arithmetic (EMA, lambdas, backprop loss, weighted seed) as two launches (`eben_balance`,
    #: `eben_weighted_sum`) instead of ~35 one-element torch kernels in front of the generator backward; same values bit for bit
    fused_balancing: bool = os.environ.get("EBEN_FUSED_BALANCING", "1") != "0"

    #: the balancing norms ||dL_i / d last_conv.weight|| of the engine step from the seeds in one pass (`eben_last_conv_norms`) instead of
    #: a tanh-backward + weight-gradient + norm chain per loss through autograd; same values up to the order of the fp32 sums
    fused_norms: bool = os.environ.get("EBEN_FUSED_NORMS", "1") != "0"

    def _update_lambdas_fused(self, atomic_norms, losses):
        """``_update_lambdas`` + the backprop loss on the device in one launch; the state stays readable as ``atomic_norms_old``."""
        import ctypes

        from .._lib import check, load, ptr, stream

        n = len(atomic_norms)
        dev = atomic_norms[0].device
        bal = getattr(self, "_bal", None)
        if bal is None or bal["n"] != n or bal["old"].device != dev:
            bal = self._bal = {"n": n, "old": torch.zeros(n, dtype=torch.float32, device=dev), "lam": None}
        old = self.atomic_norms_old
        if old is not None and len(old) == n and any(
                not torch.is_tensor(t) or t.device != dev or t.data_ptr() != bal["old"].data_ptr() + 4 * i for i, t in enumerate(old)):
            # the state was written by someone else since the last call (another step path, a restore, the user): `atomic_norms_old`
            # is the one state -- the device buffer follows it
            bal["old"].copy_(torch.stack([torch.as_tensor(t, dtype=torch.float32).to(dev).reshape(()) for t in old]))
        init = self.atomic_norms_old is None or self.dynamic_loss_balancing == "simple"
        out = torch.empty(n + 1, dtype=torch.float32, device=dev)   # fresh per step: last_lambdas of earlier steps stay what they were
        norms = [t.contiguous() for t in atomic_norms]
        ls = [t.to(torch.float32).contiguous() for t in losses]
        check(load().eben_balance((ctypes.c_void_p * n)(*[ptr(t) for t in norms]), (ctypes.c_void_p * n)(*[ptr(t) for t in ls]), n,
                                  ptr(bal["old"]), 1 if init else 0, 1 if self.dynamic_loss_balancing == "ema" else 0, float(self.beta_ema),
                                  float(1 - self.beta_ema), ptr(out), ptr(out[n:]), stream()), "balance")
        bal["lam"] = out[:n]
        self.atomic_norms_old = [bal["old"][i] for i in range(n)]
        lambdas = [out[i] for i in range(n)]
        self.last_norms, self.last_lambdas = atomic_norms, lambdas
        return lambdas, out[n]

    def dynamically_balance_losses(self, atomic_losses: Dict[str, torch.Tensor]) -> Dict[str, torch.Tensor]:
        """eben.py:222-240: one partial backward per loss down to the last generator layer, then ``_update_lambdas``."""
        leaf = self.generator.last_conv.weight
        norms = [torch.norm(torch.autograd.grad(loss, leaf, retain_graph=True)[0]).detach() for loss in atomic_losses.values()]
        return {name: loss * lam for (name, loss), lam in zip(atomic_losses.items(), self._update_lambdas(norms))}

    def on_test_end(self) -> None:
        """eben.py:176-181: upload the generator when asked to (needs Lightning's trainer / datamodule and network access)."""
        if self.push_to_hub_after_testing:
            self.generator.push_to_hub(f"Cnam-LMSSC/EBEN_{self.trainer.datamodule.sensor}",
                                       commit_message=f"Upload EBENGenerator after {self.trainer.current_epoch} epochs")
