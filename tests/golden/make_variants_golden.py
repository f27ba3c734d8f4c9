"""Generate tests/golden/eben_variants_golden.npz: the REFERENCE modules replayed through the reference's OTHER legal
``EBENLightningModule`` configurations (``vibravox/lightning_modules/eben.py:67-76, 118, 194-211, 229-235``):

  none      dynamic_loss_balancing = None              (eben.py:107-108 skipped: plain sum of the atomic losses)
  simple    dynamic_loss_balancing = "simple"          (eben.py:229-231: the norms' state is overwritten every step, no EMA)
  ratio0    update_discriminator_ratio = 0             (eben.py:118: the draw never passes; the discriminator never moves)
  ratio05   update_discriminator_ratio = 0.5, 3 steps  (the ``torch.rand(1)`` sequence of a fixed CPU seed: update / skip / update)
  l1time    reconstructive_loss_time_fn = L1Loss()     (eben.py:199-202)
  fm_only   adversarial_loss_fn = None                 (eben.py:203-211, 212-219: no discriminator term table -> no draw, no update)
  adv_only  feature_matching_loss_fn = None            (one discriminator forward in the generator phase)

Same recipe as ``make_golden.py`` (whose fixture this file leaves untouched): runs only in the build container, imports the reference
read-only, formula weights and clips, a Lightning-free replay of ``training_step`` that calls the reference's own modules in the
reference's own order.  Per step: every logged scalar, the balancing norms / lambdas where balancing is on, a summary of ``enhanced``,
whether the discriminator was updated; after the last step the per-parameter (sum, L2) checksums of both networks.  While generating,
``oracle.eben_oracle.OracleTrainer`` (the CPU restatement the GPU tests compare with at other sizes) runs the same configuration with
the same seed; the worst relative differences are stored under ``check:*``.

Usage:  python tests/golden/make_variants_golden.py
"""
from __future__ import annotations

import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)

from formula import flatten_summary, formula_audio  # noqa: E402
from make_golden import import_reference, load_formula, maxabs  # noqa: E402
from oracle import eben_oracle as O  # noqa: E402

#: name -> (constructor arguments of the replay / OracleTrainer, number of steps).  tests/test_variants.py reads the same table.
VARIANTS = {
    "none": (dict(balancing=None), 2),
    "simple": (dict(balancing="simple"), 2),
    "ratio0": (dict(ratio=0.0), 2),
    "ratio05": (dict(ratio=0.5), 3),
    "l1time": (dict(time_loss="l1"), 2),
    "fm_only": (dict(use_adv=False), 2),
    "adv_only": (dict(use_fm=False), 2),
}
#: the CPU seed set right before a variant's first step (the only CPU RNG use of a step is the draw of eben.py:118)
SEED = 7
#: ratio05 wants update / skip / update out of its three draws: the first seed >= SEED whose draws do that
RATIO05_PATTERN = (True, False, True)


def seed_for(name: str) -> int:
    if name != "ratio05":
        return SEED
    s = SEED
    while True:
        torch.manual_seed(s)
        if tuple(bool(torch.rand(1) < 0.5) for _ in RATIO05_PATTERN) == RATIO05_PATTERN:
            return s
        s += 1


def variant_batches(name: str, steps: int):
    return [(formula_audio(f"var/{name}/step{i}/bc", 2, 8200), formula_audio(f"var/{name}/step{i}/air", 2, 8200)) for i in range(steps)]


def reference_replay(R, gen, disc, batches, balancing="ema", beta_ema=0.9, ratio=1.0, time_loss=None, use_fm=True, use_adv=True):
    """eben.py:82-130 / 184-240 over the reference modules, Lightning's ``toggle_optimizer`` as requires_grad toggling."""
    fm = R["FM"]() if use_fm else None
    hinge = R["HINGE"]() if use_adv else None
    l1 = torch.nn.L1Loss() if time_loss == "l1" else None
    g_opt = torch.optim.Adam(gen.parameters(), lr=3e-4, betas=(0.5, 0.9))
    d_opt = torch.optim.Adam(disc.parameters(), lr=3e-4, betas=(0.5, 0.9))
    norms_old = None
    all_logs = []
    for corrupted, reference in batches:
        logs = {}
        corrupted = gen.cut_to_valid_length(corrupted)
        reference = gen.cut_to_valid_length(reference)
        for p in disc.parameters():
            p.requires_grad_(False)
        enhanced, dec_enh = gen(corrupted)
        dec_ref = gen.pqmf.forward(reference, "analysis")
        # compute_atomic_losses("generator"), eben.py:194-211
        losses = {}
        if l1 is not None:
            losses["reconstructive_loss_temp"] = l1(enhanced, reference)
        if fm is not None or hinge is not None:
            e_enh = disc(bands=dec_enh, audio=enhanced)
            if fm is not None:
                e_ref = disc(bands=dec_ref, audio=reference)
                losses["feature_matching_loss"] = fm(e_enh, e_ref)
            if hinge is not None:
                losses["adv_loss_gen"] = hinge(embeddings=e_enh, target=1)
        for k, v in losses.items():
            logs[f"train/generator/{k}"] = v.detach().clone()
        if balancing is not None:   # dynamically_balance_losses, eben.py:222-240
            leaf = gen.last_conv.weight
            norms = [torch.norm(torch.autograd.grad(l, leaf, retain_graph=True)[0]).detach() for l in losses.values()]
            if norms_old is None or balancing == "simple":
                norms_old = norms
            if balancing == "ema":
                norms_old = [beta_ema * o + (1 - beta_ema) * n for o, n in zip(norms_old, norms)]
            lambdas = [torch.clamp(1 / (n + 1e-4), min=0.0, max=1e4) for n in norms_old]
            for k, lam in zip(losses.keys(), lambdas):
                losses[k] *= lam
            logs["balancing/norms"] = torch.stack(norms)
            logs["balancing/lambdas"] = torch.stack(lambdas)
        total = sum(losses.values())
        logs["train/generator/backprop_loss"] = total.detach().clone()
        total.backward()
        g_opt.step()
        g_opt.zero_grad()
        for p in disc.parameters():
            p.requires_grad_(True)
        for p in gen.parameters():
            p.requires_grad_(False)
        # compute_atomic_losses("discriminator"), eben.py:212-219, and the gate of :118
        d_losses = {}
        if hinge is not None:
            e_enh = disc(bands=dec_enh.detach(), audio=enhanced.detach())
            e_ref = disc(bands=dec_ref, audio=reference)
            d_losses["real_loss"] = hinge(embeddings=e_ref, target=1)
            d_losses["fake_loss"] = hinge(embeddings=e_enh, target=-1)
        updated = bool(d_losses) and bool(torch.rand(1) < ratio)
        if updated:
            for k, v in d_losses.items():
                logs[f"train/discriminator/{k}"] = v.detach().clone()
            dtot = d_losses["real_loss"] + d_losses["fake_loss"]
            logs["train/discriminator/backprop_loss"] = dtot.detach().clone()
            dtot.backward()
            d_opt.step()
            d_opt.zero_grad()
        for n_, p in gen.named_parameters():
            p.requires_grad_(not n_.startswith("pqmf."))
        logs["enhanced"] = enhanced.detach()
        logs["updated"] = torch.tensor(float(updated))
        all_logs.append(logs)
    return all_logs


def oracle_kwargs(cfg: dict) -> dict:
    """The replay's arguments under OracleTrainer's names."""
    return dict(balancing=cfg.get("balancing", "ema"), update_discriminator_ratio=cfg.get("ratio", 1.0), time_loss=cfg.get("time_loss"),
                use_feature_matching=cfg.get("use_fm", True), use_adversarial=cfg.get("use_adv", True))


def main():
    torch.manual_seed(0)
    torch.set_num_threads(8)
    R = import_reference()
    out = {}
    for name, (cfg, steps) in VARIANTS.items():
        gen = R["G"](m=4, n=32, p=2)
        load_formula(gen, "G2")
        disc = R["D"](q=4, min_channels=24)
        load_formula(disc, "D")
        g_sd0 = {k: v.detach().clone() for k, v in gen.state_dict().items()}
        d_sd0 = {k: v.detach().clone() for k, v in disc.state_dict().items()}
        batches = variant_batches(name, steps)
        seed = seed_for(name)
        out[f"var/{name}/seed"] = np.array(seed)
        torch.manual_seed(seed)
        ref_logs = reference_replay(R, gen, disc, batches, **cfg)
        trainer = O.OracleTrainer(g_sd0, d_sd0, p=2, q=4, use_mrstft=False, **oracle_kwargs(cfg))
        torch.manual_seed(seed)
        ora_logs = [trainer.step(c, r) for c, r in batches]
        worst = 0.0
        for i, (rl, ol) in enumerate(zip(ref_logs, ora_logs)):
            updated = bool(rl["updated"])
            assert updated == ("train/discriminator/backprop_loss" in ol), (name, i)
            assert set(rl) - {"updated"} == set(ol), (name, i, sorted(rl), sorted(ol))
            for k, v in rl.items():
                if k == "enhanced":
                    flatten_summary(f"var/{name}/step{i}/enhanced", v, out)
                else:
                    out[f"var/{name}/step{i}/{k}"] = v.double().numpy()
                if k != "updated":
                    worst = max(worst, maxabs(v, ol[k]) / (v.abs().max().item() + 1e-30))
        out[f"check:var/{name}/logs_rel"] = np.array(worst)
        worst = 0.0
        for k, v in gen.state_dict().items():
            if k.startswith("pqmf."):
                continue
            out[f"var/{name}/post/G/{k}"] = np.array([v.double().sum().item(), v.double().norm().item()])
            worst = max(worst, maxabs(v, trainer.g[k]))
        moved = 0.0
        for k, v in disc.state_dict().items():
            out[f"var/{name}/post/D/{k}"] = np.array([v.double().sum().item(), v.double().norm().item()])
            worst = max(worst, maxabs(v, trainer.d[k]))
            moved = max(moved, maxabs(v, d_sd0[k]))
        out[f"check:var/{name}/post_adam_maxabs"] = np.array(worst)
        out[f"var/{name}/disc_moved"] = np.array(moved)
        pattern = [bool(l["updated"]) for l in ref_logs]
        print(f"{name:9s} seed {seed} updates {pattern} logs rel {float(out[f'check:var/{name}/logs_rel']):.2e} "
              f"post-Adam maxabs {worst:.2e} discriminator moved {moved:.2e}")
    path = os.path.join(HERE, "eben_variants_golden.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes,", len(out), "entries")


if __name__ == "__main__":
    main()
