"""CPU ORACLE for the EBEN hot path -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
``bench.py`` may import this file.  The product (``vibravox_amd``) never routes
through it: its modules raise when the HIP library is missing.

This is a plain PyTorch fp32 *functional* restatement of the reference
algorithm: every function takes a flat ``{name: tensor}`` state dict using the
reference's own ``state_dict`` key names, so reference checkpoints, the
reference modules' weights and the product's weights all drive it unchanged.
File:line citations are relative to ``/root/reference``.

Parity status
-------------
* PQMF, generator, discriminators, feature/hinge loss, balancing, train step:
  pinned -- ``tests/golden/make_golden.py`` imports the reference modules in
  the build container and freezes their outputs on formula-generated weights;
  ``tests/test_oracle_golden.py`` checks this file against those fixtures.
* ``mrstft_loss``: **parity unpinned**.  The arithmetic lives in the
  third-party ``auraloss`` package (unpinned in pyproject.toml:21, 0.4.0 was
  current at the reference date; not installed here, not vendored).  It is
  restated from the published 0.4.0 semantics (per-item spectral convergence)
  and checked only against an independent numpy rfft restatement.
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional, Sequence, Tuple

import torch
import torch.nn.functional as F

Tensor = torch.Tensor
SD = Dict[str, Tensor]

# --------------------------------------------------------------------------
# PQMF bank design -- vibravox/torch_modules/dsp/pqmf.py:66-180
# --------------------------------------------------------------------------


def _prototype(cutoff, n: int, beta: float) -> Tensor:
    """Kaiser(beta) * sinc low-pass prototype (pqmf.py:66-91).

    The Kaiser window is evaluated in float32 (torch default dtype) and only *held* in a
    float64 buffer (pqmf.py:77-80), so the product is float32*float64 -> float32.
    """
    win = torch.kaiser_window(n, periodic=False, beta=beta).double()
    centred = torch.arange(n) - (n - 1) / 2
    lowpass = cutoff * torch.special.sinc(cutoff * centred)  # float32
    out = torch.ones(1, 1, n)
    out[0, 0, :] = lowpass * win
    return out


def pqmf_cutoff(m: int, n: int, beta: float = 9) -> float:
    """Lin & Vaidyanathan cutoff search: 5 LBFGS(strong_wolfe) outer steps (pqmf.py:93-140)."""

    def phi(c):
        proto = _prototype(c, n, beta)
        padded = F.pad(proto, (n // 2, n // 2))
        acorr = F.conv1d(padded, proto)
        acorr[..., n // 2] = 0
        worst = acorr[..., :: 2 * m].abs().max()
        off = 1 / (4 * m) if abs(c - 1 / (2 * m)) > 1 / (4 * m) else 0
        return worst + off

    c = torch.ones(1) / (2 * m)
    c.requires_grad = True
    opt = torch.optim.LBFGS([c], line_search_fn="strong_wolfe")
    for _ in range(5):
        opt.zero_grad()
        phi(c).backward()
        opt.step(lambda: phi(c))
    return c.item()


def pqmf_bank(m: int, n: int, beta: float = 9, cutoff: Optional[float] = None) -> Tuple[Tensor, Tensor, float]:
    """Cosine-modulated analysis / synthesis banks, each (m,1,n) float32 (pqmf.py:142-180)."""
    if cutoff is None:
        cutoff = pqmf_cutoff(m, n, beta)
    proto = _prototype(cutoff, n, beta).squeeze()
    ana = torch.zeros(m, 1, n)
    syn = torch.zeros(m, 1, n)
    centred = torch.arange(n) - (n - 1) / 2
    for k in range(m):
        arg = (2 * k + 1) * math.pi / 2 / m * centred
        sgn = (-1) ** k * math.pi / 4
        ana[k, 0, :] = 2 * torch.flip(proto * torch.cos(arg + sgn), [0])
        syn[k, 0, :] = m * 2 * proto * torch.cos(arg - sgn)
    return ana, syn, cutoff


def pqmf_analysis(x: Tensor, ana: Tensor, bands: int = -1) -> Tensor:
    """(B,1,T) -> (B,bands,(T+n-2)//m + 1): strided FIR, zero pad n-1 (pqmf.py:194-202)."""
    m, _, n = ana.shape
    w = ana if bands == -1 else ana[:bands]
    return F.conv1d(x, w, None, stride=m, padding=n - 1)


def pqmf_synthesis(b: Tensor, syn: Tensor) -> Tensor:
    """(B,m,L) -> (B,m,m*L-n): grouped transposed FIR (pqmf.py:204-213); caller sums bands."""
    m, _, n = syn.shape
    return F.conv_transpose1d(b, syn, None, stride=m, output_padding=m - 2, groups=m, padding=n - 1)


def cut_to_valid_length(x: Tensor, n: int = 32, m: int = 4) -> Tensor:
    """eben_generator.py:215-222: T -> T - (T+n) % (2*4*8*m)."""
    t = x.shape[2]
    return x.narrow(2, 0, t - (t + n) % (2 * 4 * 8 * m))


# --------------------------------------------------------------------------
# building blocks
# --------------------------------------------------------------------------


def weight_norm(g: Tensor, v: Tensor) -> Tensor:
    """w = g * v / ||v||, norm over every dim but 0 (torch_modules/utils.py:4-9 ->
    torch.nn.utils.parametrizations.weight_norm, dim=0; for ConvTranspose1d dim 0 is Cin)."""
    nrm = v.reshape(v.shape[0], -1).norm(dim=1).reshape(-1, *([1] * (v.dim() - 1)))
    return v * (g / nrm)


def _wn(sd: SD, prefix: str) -> Tensor:
    return weight_norm(sd[prefix + ".parametrizations.weight.original0"], sd[prefix + ".parametrizations.weight.original1"])


def _conv_reflect(x: Tensor, w: Tensor, stride: int = 1, dilation: int = 1, pad: Optional[Tuple[int, int]] = None) -> Tensor:
    """Conv1d with padding_mode='reflect'; pad=None means padding='same' (stride 1)."""
    if pad is None:
        total = dilation * (w.shape[-1] - 1)
        pad = (total // 2, total - total // 2)
    if pad[0] or pad[1]:
        x = F.pad(x, pad, mode="reflect")
    return F.conv1d(x, w, None, stride=stride, dilation=dilation)


def _residual_unit(sd: SD, prefix: str, x: Tensor, dilation: int) -> Tensor:
    """x + lrelu_0.01(pointwise(dilated(x))) -- eben_generator.py:287-316."""
    h = _conv_reflect(x, _wn(sd, prefix + ".dilated_conv"), dilation=dilation)
    h = F.conv1d(h, _wn(sd, prefix + ".pointwise_conv"))
    return x + F.leaky_relu(h, 0.01)


def conv_layer(x: Tensor, v: Tensor, g: Optional[Tensor], bias: Optional[Tensor], *, stride=1, dilation=1, groups=1,
               pad_l=0, pad_r=0, reflect=False, transposed=False, output_padding=0, in_slope=1.0, out_slope=1.0) -> Tensor:
    """Reference semantics of one product conv layer (vibravox_amd.ops.conv_layer): used by the
    per-op parity tests.  lrelu_out(conv(lrelu_in(x); weight_norm(g, v)) + bias)."""
    w = weight_norm(g, v) if g is not None else v
    h = F.leaky_relu(x, in_slope) if in_slope != 1.0 else x
    if transposed:
        y = F.conv_transpose1d(h, w, bias, stride=stride, padding=pad_l, output_padding=output_padding, groups=groups, dilation=dilation)
    else:
        if pad_l or pad_r:
            h = F.pad(h, (pad_l, pad_r), mode="reflect" if reflect else "constant")
        y = F.conv1d(h, w, bias, stride=stride, dilation=dilation, groups=groups)
    return F.leaky_relu(y, out_slope) if out_slope != 1.0 else y


# --------------------------------------------------------------------------
# generator -- eben_generator.py:93-213
# --------------------------------------------------------------------------

ENC_STRIDES = (2, 4, 8)
DEC_STRIDES = (8, 4, 2)
RU_DILATIONS = (1, 3, 9)


def generator_forward(sd: SD, x: Tensor, p: int) -> Tuple[Tensor, Tensor]:
    """EBENGenerator.forward (eben_generator.py:168-213) -> (enhanced (B,1,T), bands (B,M,L0))."""
    ana, syn = sd["pqmf.analysis_weights"], sd["pqmf.synthesis_weights"]
    m = ana.shape[0]
    nl = lambda t: F.leaky_relu(t, 0.01)

    first_bands = pqmf_analysis(x, ana, bands=p)
    h = _conv_reflect(first_bands, sd["first_conv.weight"])

    skips = []
    for i, s in enumerate(ENC_STRIDES):
        h = nl(h)
        for j, d in enumerate(RU_DILATIONS):
            h = _residual_unit(sd, f"encoder_blocks.{i}.residuals.{j}", h, d)
        h = _conv_reflect(h, _wn(sd, f"encoder_blocks.{i}.conv"), stride=s, pad=(s - 1, s - 1))
        skips.append(h)

    h = nl(h)
    h = nl(_conv_reflect(h, _wn(sd, "latent_conv.1")))
    h = nl(_conv_reflect(h, _wn(sd, "latent_conv.3")))

    for i, s in enumerate(DEC_STRIDES):
        h = h + skips[2 - i]
        h = F.conv_transpose1d(h, _wn(sd, f"decoder_blocks.{i}.conv_trans"), None, stride=s, padding=s // 2)
        h = nl(h)
        for j, d in enumerate(RU_DILATIONS):
            h = _residual_unit(sd, f"decoder_blocks.{i}.residuals.{j}", h, d)

    h = _conv_reflect(h, sd["last_conv.weight"])
    b, _, t = first_bands.shape
    lifted = torch.cat((first_bands, torch.zeros(b, m - p, t, dtype=first_bands.dtype)), dim=1)
    bands = torch.tanh(h + lifted)
    enhanced = pqmf_synthesis(bands, syn).sum(1, keepdim=True)
    return enhanced, bands


# --------------------------------------------------------------------------
# discriminators -- eben_discriminator.py:10-163, melgan_discriminator.py:76-169
# --------------------------------------------------------------------------


def _disc_keys(n_layers: int) -> List[str]:
    return [f"{i}.1" if i == 0 else (f"{i}.0" if i < n_layers - 1 else f"{i}") for i in range(n_layers)]


def pqmf_disc_forward(sd: SD, prefix: str, bands: Tensor, dilation: int, q: int) -> List[Tensor]:
    """DiscriminatorEBEN.forward (eben_discriminator.py:159-163; layers :66-157)."""
    keys = _disc_keys(8)
    embs = [bands]
    h = F.pad(bands, (1, 1), mode="reflect")
    specs = [(3, 1, 1)] + [(7, 2, 3)] * 5 + [(5, 1, 2)]
    for key, (_, s, pd) in zip(keys[:7], specs):
        name = f"{prefix}.discriminator.{key}"
        h = F.conv1d(h, _wn(sd, name), sd[name + ".bias"], stride=s, padding=pd, dilation=dilation, groups=q)
        h = F.leaky_relu(h, 0.2)
        embs.append(h)
    name = f"{prefix}.discriminator.{keys[7]}"
    embs.append(F.conv1d(h, _wn(sd, name), sd[name + ".bias"], padding=1))
    return embs


def melgan_disc_forward(sd: SD, prefix: str, audio: Tensor) -> List[Tensor]:
    """DiscriminatorMelGAN.forward (melgan_discriminator.py:158-169; layers :89-156)."""
    keys = _disc_keys(7)
    embs = [audio]
    h = F.pad(audio, (7, 7), mode="reflect")
    specs = [(1, 0, 1)] + [(4, 20, 4)] * 4 + [(1, 2, 1)]
    for key, (s, pd, g) in zip(keys[:6], specs):
        name = f"{prefix}.discriminator.{key}"
        h = F.leaky_relu(F.conv1d(h, _wn(sd, name), sd[name + ".bias"], stride=s, padding=pd, groups=g), 0.2)
        embs.append(h)
    name = f"{prefix}.discriminator.{keys[6]}"
    embs.append(F.conv1d(h, _wn(sd, name), sd[name + ".bias"], padding=1))
    return embs


def discriminator_forward(sd: SD, bands: Tensor, audio: Tensor, q: int) -> List[List[Tensor]]:
    """DiscriminatorEBENMultiScales.forward (eben_discriminator.py:33-51): dilations 1,2,3 then MelGAN."""
    out = [pqmf_disc_forward(sd, f"pqmf_discriminators.{i}", bands[:, -q:, :], d, q) for i, d in enumerate((1, 2, 3))]
    out.append(melgan_disc_forward(sd, "melgan_discriminator", audio))
    return out


# --------------------------------------------------------------------------
# losses
# --------------------------------------------------------------------------


def feature_loss(emb_a: List[List[Tensor]], emb_b: List[List[Tensor]]) -> Tensor:
    """feature_loss.py:37-50.  Quirks kept: normalised by mean|a| (the *enhanced* side), and
    divided by n_scales * n_inner_layers_of_the_LAST_scale (4*6=24 although 27 terms are summed)."""
    total = 0.0
    for sa, sb in zip(emb_a, emb_b):
        for la, lb in zip(sa[1:-1], sb[1:-1]):
            total = total + (la - lb).abs().mean() / la.abs().mean()
    return total / (len(emb_a) * len(emb_a[-1][1:-1]))


def hinge_loss(emb: List[List[Tensor]], target: float) -> Tensor:
    """hinge_loss.py:35-43."""
    total = 0.0
    for scale in emb:
        total = total + F.relu(1 - target * scale[-1]).mean()
    return total / len(emb)


def a_weighting_fir(fs: float, ntaps: int = 101) -> Tensor:
    """auraloss.perceptual.FIRFilter(filter_type='aw') taps [3p-memory, auraloss 0.4.0]:
    IEC analog A-weighting prototype -> bilinear -> freqz(512) -> firls(ntaps)."""
    import numpy as np
    import scipy.signal

    f1, f2, f3, f4, a1000 = 20.598997, 107.65265, 737.86223, 12194.217, 1.9997
    num = [(2 * np.pi * f4) ** 2 * (10 ** (a1000 / 20)), 0, 0, 0, 0]
    den = np.polymul([1, 4 * np.pi * f4, (2 * np.pi * f4) ** 2], [1, 4 * np.pi * f1, (2 * np.pi * f1) ** 2])
    den = np.polymul(np.polymul(den, [1, 2 * np.pi * f3]), [1, 2 * np.pi * f2])
    b, a = scipy.signal.bilinear(num, den, fs=fs)
    w, h = scipy.signal.freqz(b, a, worN=512, fs=fs)
    taps = scipy.signal.firls(ntaps, w, abs(h), fs=fs)
    return torch.tensor(taps.astype("float32"))


def stft_mag(x: Tensor, n_fft: int, hop: int, win: int, eps: float = 1e-8) -> Tensor:
    """(N,T) -> (N, n_fft/2+1, frames): sqrt(clamp(re^2+im^2, eps)), hann(win) centred in n_fft,
    center=True reflect (auraloss STFTLoss.stft [3p-memory])."""
    spec = torch.stft(x, n_fft, hop, win, torch.hann_window(win), return_complex=True)
    return torch.sqrt(torch.clamp(spec.real ** 2 + spec.imag ** 2, min=eps))


def mrstft_loss(
    x: Tensor,
    y: Tensor,
    fft_sizes: Sequence[int] = (512, 1024, 2048),
    hop_sizes: Sequence[int] = (50, 120, 240),
    win_lengths: Sequence[int] = (240, 600, 1200),
    sample_rate: int = 16000,
    perceptual_weighting: bool = True,
    fir: Optional[Tensor] = None,
) -> Tensor:
    """auraloss.freq.MultiResolutionSTFTLoss(x=enhanced, y=reference) as configured by
    configs/lightning_module/loss_module/multi_stft.yaml:1-18 (call site eben.py:195-198).
    PARITY UNPINNED (third-party, see module docstring).  Per resolution:
    A-weighting FIR on both -> |STFT| -> mean_items(||Y-X||_F / ||Y||_F) + L1(log X, log Y)."""
    b, c, t = x.shape
    if perceptual_weighting:
        if fir is None:
            fir = a_weighting_fir(sample_rate)
        k = fir.view(1, 1, -1)
        x = F.conv1d(x.reshape(b * c, 1, t), k, padding=k.shape[-1] // 2).view(b, c, -1)
        y = F.conv1d(y.reshape(b * c, 1, t), k, padding=k.shape[-1] // 2).view(b, c, -1)
    total = 0.0
    for n_fft, hop, win in zip(fft_sizes, hop_sizes, win_lengths):
        xm = stft_mag(x.reshape(-1, x.shape[-1]), n_fft, hop, win)
        ym = stft_mag(y.reshape(-1, y.shape[-1]), n_fft, hop, win)
        sc = (torch.norm(ym - xm, p="fro", dim=[-1, -2]) / torch.norm(ym, p="fro", dim=[-1, -2])).mean()
        lg = (torch.log(xm) - torch.log(ym)).abs().mean()
        total = total + sc + lg
    return total / len(fft_sizes)


# --------------------------------------------------------------------------
# the train step -- vibravox/lightning_modules/eben.py:82-130, 184-240
# --------------------------------------------------------------------------


class OracleTrainer:
    """Lightning-free replay of EBENLightningModule.training_step with the defaults of
    configs/lightning_module/eben.yaml (MRSTFT + feature matching + hinge, EMA balancing,
    Adam lr 3e-4 betas (0.5, 0.9)).  ``as_executed=True`` follows the reference's order
    exactly (4 discriminator forwards, 3 balancing partial backwards + the final one)."""

    def __init__(self, g_sd: SD, d_sd: SD, p: int = 2, q: int = 4, lr: float = 3e-4, betas=(0.5, 0.9),
                 balancing: Optional[str] = "ema", beta_ema: float = 0.9, sample_rate: int = 16000,
                 use_mrstft: bool = True, update_discriminator_ratio: float = 1.0, time_loss: Optional[str] = None,
                 use_feature_matching: bool = True, use_adversarial: bool = True):
        """The optional arguments are the reference's other legal configurations (eben.py:67-76, 194-211):
        ``balancing`` None / "simple" / "ema"; ``update_discriminator_ratio`` in [0, 1] (the ``torch.rand(1)`` draw of :118,
        taken from torch's global CPU generator exactly where the reference takes it); ``time_loss`` "l1" =
        ``reconstructive_loss_time_fn = torch.nn.L1Loss()``; feature-matching-only / adversarial-only."""
        assert balancing in {None, "simple", "ema"} and 0 <= update_discriminator_ratio <= 1 and time_loss in {None, "l1"}
        self.g = {k: v.clone().requires_grad_(v.dtype.is_floating_point and not k.startswith("pqmf.")) for k, v in g_sd.items()}
        self.d = {k: v.clone().requires_grad_(True) for k, v in d_sd.items()}
        self.p, self.q = p, q
        self.g_params = [v for v in self.g.values() if v.requires_grad]
        self.d_params = list(self.d.values())
        self.g_opt = torch.optim.Adam(self.g_params, lr=lr, betas=betas)
        self.d_opt = torch.optim.Adam(self.d_params, lr=lr, betas=betas)
        self.balancing, self.beta_ema = balancing, beta_ema
        self.norms_old: Optional[List[Tensor]] = None
        self.fir = a_weighting_fir(sample_rate) if use_mrstft else None
        self.use_mrstft = use_mrstft
        self.sample_rate = sample_rate
        self.ratio, self.time_loss = update_discriminator_ratio, time_loss
        self.use_fm, self.use_adv = use_feature_matching, use_adversarial

    def _balance(self, losses: Dict[str, Tensor]) -> Tuple[Dict[str, Tensor], List[Tensor], List[Tensor]]:
        """eben.py:222-240 incl. the first-call quirk (init with current norms, then EMA anyway)."""
        leaf = self.g["last_conv.weight"]
        norms = [torch.autograd.grad(v, leaf, retain_graph=True)[0].norm().detach() for v in losses.values()]
        if self.norms_old is None or self.balancing == "simple":
            self.norms_old = norms
        if self.balancing == "ema":
            self.norms_old = [self.beta_ema * o + (1 - self.beta_ema) * n for o, n in zip(self.norms_old, norms)]
        lambdas = [torch.clamp(1 / (n + 1e-4), min=0.0, max=1e4) for n in self.norms_old]
        return {k: v * l for (k, v), l in zip(losses.items(), lambdas)}, norms, lambdas

    def step(self, corrupted: Tensor, reference: Tensor) -> Dict[str, Tensor]:
        logs: Dict[str, Tensor] = {}
        corrupted = cut_to_valid_length(corrupted)
        reference = cut_to_valid_length(reference)

        # ---- generator phase (eben.py:96-111); D frozen by toggle_optimizer
        for t in self.d_params:
            t.requires_grad_(False)
        enhanced, bands_enh = generator_forward(self.g, corrupted, self.p)
        bands_ref = pqmf_analysis(reference, self.g["pqmf.analysis_weights"])
        losses: Dict[str, Tensor] = {}
        if self.use_mrstft:
            losses["reconstructive_loss_freq"] = mrstft_loss(enhanced, reference, sample_rate=self.sample_rate, fir=self.fir)
        if self.time_loss == "l1":   # eben.py:199-202 with torch.nn.L1Loss
            losses["reconstructive_loss_temp"] = (enhanced - reference).abs().mean()
        if self.use_fm or self.use_adv:   # eben.py:203-211
            emb_enh = discriminator_forward(self.d, bands_enh, enhanced, self.q)
            if self.use_fm:
                emb_ref = discriminator_forward(self.d, bands_ref, reference, self.q)
                losses["feature_matching_loss"] = feature_loss(emb_enh, emb_ref)
            if self.use_adv:
                losses["adv_loss_gen"] = hinge_loss(emb_enh, 1)
        for k, v in losses.items():
            logs[f"train/generator/{k}"] = v.detach().clone()
        if self.balancing is not None:
            losses, norms, lambdas = self._balance(losses)
            logs["balancing/norms"] = torch.stack(norms)
            logs["balancing/lambdas"] = torch.stack([l.detach() for l in lambdas])
        total = sum(losses.values())
        logs["train/generator/backprop_loss"] = total.detach().clone()
        total.backward()
        self.g_opt.step()
        self.g_opt.zero_grad()
        for t in self.d_params:
            t.requires_grad_(True)

        # ---- discriminator phase (eben.py:114-128); G frozen.  Without an adversarial loss the term table is empty and the
        # `and` of :118 short-circuits: no draw is taken.
        logs["enhanced"] = enhanced.detach()
        if not self.use_adv:
            return logs
        emb_enh = discriminator_forward(self.d, bands_enh.detach(), enhanced.detach(), self.q)
        emb_ref = discriminator_forward(self.d, bands_ref, reference, self.q)
        real = hinge_loss(emb_ref, 1)
        fake = hinge_loss(emb_enh, -1)
        if not bool(torch.rand(1) < self.ratio):
            return logs
        logs["train/discriminator/real_loss"] = real.detach().clone()
        logs["train/discriminator/fake_loss"] = fake.detach().clone()
        d_total = real + fake
        logs["train/discriminator/backprop_loss"] = d_total.detach().clone()
        d_total.backward()
        self.d_opt.step()
        self.d_opt.zero_grad()
        return logs
