"""Generate tests/golden/eben_golden.npz by running the REFERENCE modules.

Runs only in the build container (needs /root/reference).  The reference is
imported read-only; ``torchaudio`` (absent here, imported at module top by
melgan_discriminator.py:12 but never used on the EBEN path) is stubbed in a
temp dir.  Weights and inputs are the closed-form tensors of ``formula.py``.

What is frozen (all float64 summaries: shape / sum / L2 / 16 strided probes):
  * PQMF design: banks for (M,N)=(4,32), cutoff ratios for (4,32),(8,64),(32,1024)
  * cut_to_valid_length table
  * generator forward (p=2 and p=1), its gradients per parameter
  * discriminator embeddings (4 scales), FM / hinge(+-1) losses, gradients
  * a Lightning-free replay of EBENLightningModule.training_step (eben.py:82-130,
    184-240) over the reference modules for 2 steps *without* the third-party
    MRSTFT term (reconstructive_loss_freq_fn=None is a legal reference config,
    eben.py:194): logged scalars, balancing norms / lambdas, post-Adam checksums.
While generating, the CPU oracle is checked against the reference on the full
tensors; the max-abs differences are printed and stored under ``check:*``.

Usage:  python tests/golden/make_golden.py
"""
from __future__ import annotations

import os
import sys
import tempfile

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)

from formula import flatten_summary, formula_audio, formula_state_dict, iter_embeddings  # noqa: E402
from oracle import eben_oracle as O  # noqa: E402


def import_reference():
    stub = tempfile.mkdtemp(prefix="ta_stub_")
    os.makedirs(os.path.join(stub, "torchaudio"))
    open(os.path.join(stub, "torchaudio", "__init__.py"), "w").close()
    with open(os.path.join(stub, "torchaudio", "transforms.py"), "w") as f:
        f.write("import torch\nclass Resample(torch.nn.Module):\n    def __init__(s,*a,**k):\n        raise NotImplementedError\n")
    sys.path.insert(0, stub)
    sys.path.insert(0, "/root/reference")
    from vibravox.torch_modules.dnn.eben_discriminator import DiscriminatorEBENMultiScales
    from vibravox.torch_modules.dnn.eben_generator import EBENGenerator
    from vibravox.torch_modules.dsp.pqmf import PseudoQMFBanks
    from vibravox.torch_modules.losses.feature_loss import FeatureLossForDiscriminatorMelganMultiScales
    from vibravox.torch_modules.losses.hinge_loss import HingeLossForDiscriminatorMelganMultiScales

    return dict(G=EBENGenerator, D=DiscriminatorEBENMultiScales, PQMF=PseudoQMFBanks,
                FM=FeatureLossForDiscriminatorMelganMultiScales, HINGE=HingeLossForDiscriminatorMelganMultiScales)


def load_formula(module: torch.nn.Module, tag: str):
    shapes = {k: tuple(v.shape) for k, v in module.state_dict().items()}
    sd = formula_state_dict(shapes, tag)
    missing, unexpected = module.load_state_dict(sd, strict=False)
    assert not unexpected and all(k.startswith("pqmf.") for k in missing), (missing, unexpected)
    return shapes


def maxabs(a, b):
    return float((a.detach().double() - b.detach().double()).abs().max())


def rel_l2(a, b):
    """||a-b|| / ||a||: robust to the isolated LeakyReLU-mask / sign(a-b) flips that fp32
    rounding noise causes (the reference's own fp32 and fp64 gradients differ by several
    percent max-abs on the small deep layers for that reason)."""
    a, b = a.detach().double(), b.detach().double()
    return float((a - b).norm() / (a.norm() + 1e-30))


def reference_replay(R, gen, disc, batches, with_balancing="ema", beta_ema=0.9):
    """Lightning-free replay of eben.py:82-130 / 184-240 over the reference modules."""
    fm, hinge = R["FM"](), R["HINGE"]()
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
        losses = {}
        e_enh = disc(bands=dec_enh, audio=enhanced)
        e_ref = disc(bands=dec_ref, audio=reference)
        losses["feature_matching_loss"] = fm(e_enh, e_ref)
        losses["adv_loss_gen"] = hinge(embeddings=e_enh, target=1)
        for k, v in losses.items():
            logs[f"train/generator/{k}"] = v.detach().clone()
        leaf = gen.last_conv.weight
        norms = [torch.norm(torch.autograd.grad(l, leaf, retain_graph=True)[0]).detach() for l in losses.values()]
        if norms_old is None:
            norms_old = norms
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
        e_enh = disc(bands=dec_enh.detach(), audio=enhanced.detach())
        e_ref = disc(bands=dec_ref, audio=reference)
        real = hinge(embeddings=e_ref, target=1)
        fake = hinge(embeddings=e_enh, target=-1)
        assert torch.rand(1) < 1.0  # eben.py:118 draw, ratio = 1
        logs["train/discriminator/real_loss"] = real.detach().clone()
        logs["train/discriminator/fake_loss"] = fake.detach().clone()
        dtot = real + fake
        logs["train/discriminator/backprop_loss"] = dtot.detach().clone()
        dtot.backward()
        d_opt.step()
        d_opt.zero_grad()
        for n_, p in gen.named_parameters():
            p.requires_grad_(not n_.startswith("pqmf."))
        logs["enhanced"] = enhanced.detach()
        all_logs.append(logs)
    return all_logs


def main():
    torch.manual_seed(0)
    torch.set_num_threads(8)
    R = import_reference()
    out = {}

    # ---- PQMF design
    pq = R["PQMF"](decimation=4, kernel_size=32)
    out["pqmf/analysis_4_32"] = pq.analysis_weights.detach().numpy().copy()
    out["pqmf/synthesis_4_32"] = pq.synthesis_weights.detach().numpy().copy()
    cut = [pq._cutoff_ratio]
    for m, n in ((8, 64), (32, 1024)):
        cut.append(R["PQMF"](decimation=m, kernel_size=n)._cutoff_ratio)
    out["pqmf/cutoffs"] = np.array(cut, dtype=np.float64)
    ana, syn, c = O.pqmf_bank(4, 32)
    out["check:pqmf_bank"] = np.array(max(maxabs(ana, pq.analysis_weights), maxabs(syn, pq.synthesis_weights)))
    out["check:pqmf_cutoff"] = np.array(abs(c - pq._cutoff_ratio))
    print("pqmf bank maxabs", out["check:pqmf_bank"], "cutoff diff", out["check:pqmf_cutoff"], cut)

    # ---- cut_to_valid_length table
    gen = R["G"](m=4, n=32, p=2)
    lens = [255, 256, 480, 1000, 15679, 16000, 32000, 40000, 48009]
    out["cut/in"] = np.array(lens)
    out["cut/out"] = np.array([gen.cut_to_valid_length(torch.zeros(1, 1, L)).shape[2] for L in lens])

    # ---- generator forward + grads, p=2 (yaml default) and p=1 (reference tests' fixture)
    for p in (2, 1):
        gen = R["G"](m=4, n=32, p=p)
        load_formula(gen, f"G{p}")
        x = gen.cut_to_valid_length(formula_audio("g_in", 2, 8192))
        enh, bands = gen(x)
        sd = {k: v.detach().clone() for k, v in gen.state_dict().items()}
        sd_req = {k: v.requires_grad_(not k.startswith("pqmf.")) for k, v in sd.items()}
        o_enh, o_bands = O.generator_forward(sd_req, x, p)
        out[f"check:gen_p{p}"] = np.array(max(maxabs(enh, o_enh), maxabs(bands, o_bands)))
        print(f"generator p={p} maxabs ref-vs-oracle", out[f"check:gen_p{p}"])
        flatten_summary(f"gen{p}/enhanced", enh, out)
        flatten_summary(f"gen{p}/bands", bands, out)
        # a scalar objective exercising both outputs
        wgt_e = formula_audio("g_seed", 2, enh.shape[2], amp=1.0)
        loss = (enh * wgt_e).sum() + (bands ** 2).sum()
        loss.backward()
        o_loss = (o_enh * wgt_e).sum() + (o_bands ** 2).sum()
        o_loss.backward()
        worst = 0.0
        for k, prm in gen.named_parameters():
            if prm.grad is None:
                continue
            key = k
            out[f"gen{p}/grad_l2/{key}"] = np.array(prm.grad.double().norm().item())
            worst = max(worst, rel_l2(prm.grad, sd_req[key].grad))
        out[f"check:gen_p{p}_grad_rel"] = np.array(worst)
        print(f"generator p={p} worst relative grad diff", worst)
        if p == 2:
            # BASELINE config 1: B=4 x 16000 -> 15840, forward only
            x1 = gen.cut_to_valid_length(formula_audio("cfg1", 4, 16000))
            with torch.no_grad():
                e1, b1 = gen(x1)
            flatten_summary("cfg1/enhanced", e1, out)
            flatten_summary("cfg1/bands", b1, out)

    # ---- discriminator
    gen = R["G"](m=4, n=32, p=2)
    load_formula(gen, "G2")
    disc = R["D"](q=4, min_channels=24)
    load_formula(disc, "D")
    bands = formula_audio("d_bands", 8, 2016, amp=0.5).reshape(2, 4, 2016)
    audio = formula_audio("d_audio", 2, 4 * 2016 - 32)
    bands_b = formula_audio("d_bands_b", 8, 2016, amp=0.5).reshape(2, 4, 2016)
    audio_b = formula_audio("d_audio_b", 2, 4 * 2016 - 32)
    bands.requires_grad_(True)
    audio.requires_grad_(True)
    e_a = disc(bands=bands, audio=audio)
    with torch.no_grad():
        e_b = disc(bands=bands_b, audio=audio_b)
    dsd = {k: v.detach().clone().requires_grad_(True) for k, v in disc.state_dict().items()}
    ob, oa = bands.detach().clone().requires_grad_(True), audio.detach().clone().requires_grad_(True)
    o_a = O.discriminator_forward(dsd, ob, oa, 4)
    with torch.no_grad():
        o_b = O.discriminator_forward(dsd, bands_b, audio_b, 4)
    worst = 0.0
    for (name, t), (_, u) in zip(iter_embeddings(e_a), iter_embeddings(o_a)):
        flatten_summary(f"disc/{name}", t, out)
        worst = max(worst, maxabs(t, u))
    out["check:disc_fwd"] = np.array(worst)
    print("discriminator fwd maxabs", worst)
    fm, hinge = R["FM"](), R["HINGE"]()
    l_fm, l_hp, l_hm = fm(e_a, e_b), hinge(embeddings=e_a, target=1), hinge(embeddings=e_a, target=-1)
    out["loss/fm"], out["loss/hinge_p1"], out["loss/hinge_m1"] = (np.array(v.item(), dtype=np.float64) for v in (l_fm, l_hp, l_hm))
    o_fm, o_hp, o_hm = O.feature_loss(o_a, o_b), O.hinge_loss(o_a, 1), O.hinge_loss(o_a, -1)
    out["check:losses"] = np.array(max(abs(l_fm.item() - o_fm.item()), abs(l_hp.item() - o_hp.item()), abs(l_hm.item() - o_hm.item())))
    print("loss diffs", out["check:losses"], l_fm.item(), l_hp.item(), l_hm.item())
    (l_fm + 0.5 * l_hp + 0.25 * l_hm).backward()
    (o_fm + 0.5 * o_hp + 0.25 * o_hm).backward()
    flatten_summary("disc/grad_bands", bands.grad, out)
    flatten_summary("disc/grad_audio", audio.grad, out)
    worst = max(rel_l2(bands.grad, ob.grad), rel_l2(audio.grad, oa.grad))
    for k, prm in disc.named_parameters():
        out[f"disc/grad_l2/{k}"] = np.array(prm.grad.double().norm().item())
        worst = max(worst, rel_l2(prm.grad, dsd[k].grad))
    out["check:disc_grad_rel"] = np.array(worst)
    print("discriminator worst rel-L2 grad diff (oracle vs reference, fp32)", worst)
    # context: the reference against itself in float64 (noise floor of the discontinuous graph)
    disc64 = R["D"](q=4, min_channels=24)
    load_formula(disc64, "D")
    disc64 = disc64.double()
    b64, a64 = bands.detach().double().requires_grad_(True), audio.detach().double().requires_grad_(True)
    e64 = disc64(bands=b64, audio=a64)
    with torch.no_grad():
        e64b = disc64(bands=bands_b.double(), audio=audio_b.double())
    (fm(e64, e64b) + 0.5 * hinge(embeddings=e64, target=1) + 0.25 * hinge(embeddings=e64, target=-1)).backward()
    floor = max(rel_l2(p64.grad, p32.grad) for p64, p32 in zip(disc64.parameters(), disc.parameters()))
    out["check:disc_grad_fp64_floor"] = np.array(floor)
    print("  (reference fp32 vs reference fp64 worst rel-L2:", floor, ")")

    # ---- the state_dict contract (key order + shapes) of both networks
    for tag, mod in (("G", gen), ("D", disc)):
        sd = mod.state_dict()
        out[f"contract/{tag}/keys"] = np.array(list(sd.keys()))
        out[f"contract/{tag}/shapes"] = np.array([",".join(map(str, v.shape)) for v in sd.values()])

    # ---- two train steps without MRSTFT
    gen = R["G"](m=4, n=32, p=2)
    load_formula(gen, "G2")
    disc = R["D"](q=4, min_channels=24)
    load_formula(disc, "D")
    g_sd0 = {k: v.detach().clone() for k, v in gen.state_dict().items()}
    d_sd0 = {k: v.detach().clone() for k, v in disc.state_dict().items()}
    batches = [(formula_audio(f"step{i}/bc", 2, 8200), formula_audio(f"step{i}/air", 2, 8200)) for i in range(2)]
    ref_logs = reference_replay(R, gen, disc, batches)
    trainer = O.OracleTrainer(g_sd0, d_sd0, p=2, q=4, use_mrstft=False)
    ora_logs = [trainer.step(c, r) for c, r in batches]
    worst = 0.0
    for i, (rl, ol) in enumerate(zip(ref_logs, ora_logs)):
        for k, v in rl.items():
            if k == "enhanced":
                flatten_summary(f"step{i}/enhanced", v, out)
            else:
                out[f"step{i}/{k}"] = v.double().numpy()
            worst = max(worst, maxabs(v, ol[k]) / (v.abs().max().item() + 1e-30))
    out["check:train_logs_rel"] = np.array(worst)
    print("train-step logs worst relative diff", worst)
    worst = 0.0
    for k, v in gen.state_dict().items():
        if k.startswith("pqmf."):
            continue
        out[f"post/G/{k}"] = np.array([v.double().sum().item(), v.double().norm().item()])
        worst = max(worst, maxabs(v, trainer.g[k]))
    for k, v in disc.state_dict().items():
        out[f"post/D/{k}"] = np.array([v.double().sum().item(), v.double().norm().item()])
        worst = max(worst, maxabs(v, trainer.d[k]))
    out["check:post_adam_maxabs"] = np.array(worst)
    print("post-Adam params maxabs ref-vs-oracle", worst)

    path = os.path.join(HERE, "eben_golden.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes,", len(out), "entries")


if __name__ == "__main__":
    main()
