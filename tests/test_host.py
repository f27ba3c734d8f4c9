"""CPU (no GPU): the C-ABI library loads and exports every symbol include/eben_hip.h declares,
the drop-in contract (state_dict keys / shapes / same-seed init order, constructor asserts, helper
semantics) holds, there is NO CPU fallback, and the oracle's torch convolutions agree with the
independent plain-C restatement (oracle/conv_ref.c)."""
import ctypes
import os
import re
import subprocess

import numpy as np
import pytest
import torch

from formula import formula_tensor
from oracle import eben_oracle as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    from vibravox_amd import _lib

    if not os.path.exists(_lib.LIB_PATH):
        import __graft_entry__ as g

        g.build()
    return _lib.load()


def test_library_exports_every_declared_symbol(lib):
    from vibravox_amd import _lib

    header = open(os.path.join(ROOT, "include", "eben_hip.h")).read()
    declared = set(re.findall(r"EBEN_API\s+[\w\s\*]+?\b(eben_\w+)\s*\(", header))
    assert len(declared) >= 30
    assert declared == set(_lib.SIGNATURES), declared ^ set(_lib.SIGNATURES)
    for name in declared:
        assert hasattr(lib, name), name
    from vibravox_amd._lib import ABI_VERSION
    assert lib.eben_version() == ABI_VERSION == int(re.search(r"#define EBEN_ABI_VERSION (\d+)", header).group(1))


def test_no_kernel_of_the_library_spills(lib, tmp_path):
    """Every gfx950 kernel of libeben_hip.so runs out of registers only: the code objects' notes (llvm-readelf, no GPU needed) report a
    zero private segment (scratch) for each of them -- a spill in a hot loop is a silent 2-5x (round 1 shipped five such kernels)."""
    import shutil
    import subprocess

    llvm = "/opt/rocm/lib/llvm/bin"
    if not os.path.exists(os.path.join(llvm, "llvm-objdump")):
        pytest.skip("ROCm llvm tools not installed")
    so = shutil.copy(os.path.join(ROOT, "vibravox_amd", "lib", "libeben_hip.so"), tmp_path / "lib.so")
    subprocess.run([os.path.join(llvm, "llvm-objdump"), "--offloading", str(so)], check=True, capture_output=True, cwd=tmp_path)
    objs = [f for f in os.listdir(tmp_path) if "gfx950" in f]
    assert objs, "no gfx950 code object in the library"
    kernels, spilling = 0, []
    for f in objs:
        notes = subprocess.run([os.path.join(llvm, "llvm-readelf"), "--notes", str(tmp_path / f)], check=True, capture_output=True, text=True).stdout
        for blk in notes.split(".agpr_count:")[1:]:
            name = re.search(r"\.name:\s+(\S+)", blk)
            scratch = re.search(r"\.private_segment_fixed_size:\s+(\d+)", blk)
            if name is None or scratch is None:
                continue
            kernels += 1
            if int(scratch.group(1)) != 0:
                spilling.append((name.group(1), int(scratch.group(1))))
    assert kernels > 100, kernels
    assert not spilling, spilling


def test_descriptor_validation_without_gpu(lib):
    """Pure host logic of the ABI: bad descriptors are rejected with a negative code and a message."""
    from vibravox_amd._lib import EbenConv1dDesc

    good = EbenConv1dDesc(2, 16, 64, 1000, 250, 41, 4, 1, 4, 20, 20, 0, 0, 1.0, 0.2)
    assert lib.eben_conv1d_packed_floats(ctypes.byref(good), 0) > 0
    assert lib.eben_conv1d_packed_floats(ctypes.byref(good), 1) > 0
    nslab, rs = ctypes.c_int(0), ctypes.c_int(0)
    assert lib.eben_conv1d_bwd_dw_workspace(ctypes.byref(good), ctypes.byref(nslab), ctypes.byref(rs)) > 0
    assert rs.value == 4 * 41 + 1 and nslab.value >= 1
    bad = EbenConv1dDesc(2, 16, 64, 1000, 251, 41, 4, 1, 4, 20, 20, 0, 0, 1.0, 0.2)  # wrong l_out
    assert lib.eben_conv1d_packed_floats(ctypes.byref(bad), 0) == 0
    assert b"l_out" in lib.eben_last_error()
    bad2 = EbenConv1dDesc(2, 15, 64, 1000, 250, 41, 4, 1, 4, 20, 20, 0, 0, 1.0, 0.2)  # channels % groups
    assert lib.eben_conv1d_packed_floats(ctypes.byref(bad2), 0) == 0
    refl = EbenConv1dDesc(2, 32, 32, 100, 100, 3, 1, 9, 1, 9, 9, 1, 0, 1.0, 1.0)
    assert lib.eben_conv1d_bwd_dx_workspace(ctypes.byref(refl)) == 4 * 2 * 32 * 118


def test_kernel_family_selection_without_gpu(lib):
    """Host-side planning of the three tap-conv families at the BASELINE config-2 shapes (no launch)."""
    from vibravox_amd._lib import EbenConv1dDesc

    def gen(desc, which):
        return lib.eben_conv1d_kernel_generation(ctypes.byref(desc), which)

    melgan_l4 = EbenConv1dDesc(64, 1024, 1024, 500, 125, 41, 4, 1, 4, 20, 20, 0, 0, 1.0, 0.2)
    assert gen(melgan_l4, 0) == 2 and gen(melgan_l4, 1) == 2            # second-generation MFMA kernel
    pqmf_l1 = EbenConv1dDesc(64, 24, 48, 8000, 4000, 7, 2, 1, 4, 3, 3, 0, 0, 1.0, 0.2)
    assert gen(pqmf_l1, 0) == 3 and gen(pqmf_l1, 1) == 3                # direct kernel: 12 / 6 rows per group
    logits = EbenConv1dDesc(64, 1024, 1, 125, 125, 3, 1, 1, 1, 1, 1, 0, 0, 1.0, 1.0)
    assert gen(logits, 0) == 1                                           # single-output-channel reduction
    ru = EbenConv1dDesc(32, 64, 64, 4000, 4000, 3, 1, 9, 1, 9, 9, 1, 0, 0.01, 1.0)
    assert gen(ru, 0) == 2 and gen(ru, 1) == 2
    # the generator's strided / transposed / latent convs under the forward's fp32-grade math: their own kernel (gen_conv.hip, 5) for the
    # forward, the general tap-conv for the input gradient; the image is sized by the same plan
    X6 = 4
    enc3 = EbenConv1dDesc(32, 128, 256, 999, 125, 16, 8, 1, 1, 7, 7, 1, 0, 1.0, 1.0, X6)
    dec2 = EbenConv1dDesc(32, 128, 64, 999, 3996, 8, 4, 1, 1, 2, 0, 0, 1, 1.0, 0.01, X6)
    lat1 = EbenConv1dDesc(32, 256, 64, 125, 125, 7, 1, 1, 1, 3, 3, 1, 0, 0.01, 0.01, X6)
    for dsc in (enc3, dec2, lat1):
        assert gen(dsc, 0) == 5 and gen(dsc, 1) != 5
    # [half][32-row tile][k-step, padded to 8][piece][lane] of 16-byte units
    assert lib.eben_conv1d_packed_floats(ctypes.byref(enc3), 0) == 1 * 8 * 128 * 3 * 64 * 4
    assert lib.eben_conv1d_packed_floats(ctypes.byref(dec2), 0) == 2 * 4 * 16 * 3 * 64 * 4
    grouped = EbenConv1dDesc(32, 128, 256, 999, 125, 16, 8, 1, 4, 7, 7, 1, 0, 1.0, 1.0, X6)
    assert gen(grouped, 0) != 5                                          # groups stay with the tap-conv
    nslab, rs = ctypes.c_int(0), ctypes.c_int(0)
    assert lib.eben_conv1d_bwd_dw_workspace(ctypes.byref(melgan_l4), ctypes.byref(nslab), ctypes.byref(rs)) > 4 * 1024 * (256 * 41 + 1)
    assert rs.value == 256 * 41 + 1
    assert lib.eben_stft_loss_sums_workspace(32) == 4 * 3 * 32 * 32


def test_discriminator_engine_layout():
    """disc_engine.py sees the reference's module tree: 3 x (ReflectionPad1d(1) + 8 convs), 1 x (pad 7 + 7 convs),
    and parameter order == discriminator.parameters() order (what inject_grads relies on)."""
    from vibravox_amd.disc_engine import DiscriminatorEngine
    from vibravox_amd.torch_modules.dnn.eben_discriminator import DiscriminatorEBENMultiScales

    disc = DiscriminatorEBENMultiScales(q=4, min_channels=24)
    assert DiscriminatorEngine.supports(disc) and not DiscriminatorEngine.supports(torch.nn.Linear(2, 2))
    eng = DiscriminatorEngine(disc)
    assert [len(c.layers) for c in eng.chains] == [8, 8, 8, 7] and [c.pad for c in eng.chains] == [1, 1, 1, 7]
    seen = []
    for ch in eng.chains:
        for lay in ch.layers:
            v, g, bias = lay.params()
            assert lay.spec_lin.out_slope == 1.0 and lay.spec_lin.in_slope == 1.0 and lay.spec_lin.c_out == lay.spec.c_out
            seen += [id(bias), id(g), id(v)]
    assert seen == [id(p) for p in disc.parameters()]
    assert [lay.spec.out_slope for lay in eng.chains[0].layers] == [0.2] * 7 + [1.0]


def test_state_dict_contract_and_same_seed_init(golden):
    from vibravox_amd.torch_modules.dnn.eben_discriminator import DiscriminatorEBENMultiScales
    from vibravox_amd.torch_modules.dnn.eben_generator import EBENGenerator

    torch.manual_seed(42)
    gen, disc = EBENGenerator(m=4, n=32, p=2), DiscriminatorEBENMultiScales(q=4, min_channels=24)
    for tag, mod in (("G", gen), ("D", disc)):
        sd = mod.state_dict()
        assert list(sd.keys()) == list(golden[f"contract/{tag}/keys"])
        assert [",".join(map(str, v.shape)) for v in sd.values()] == list(golden[f"contract/{tag}/shapes"])
    assert sum(p.numel() for p in gen.parameters()) == 1_946_240
    assert sum(p.numel() for p in disc.parameters()) == 23_161_344
    assert not gen.pqmf.analysis_weights.requires_grad and gen.last_conv.weight.is_leaf and gen.last_conv.weight.requires_grad
    np.testing.assert_array_equal(gen.pqmf.analysis_weights.numpy(), golden["pqmf/analysis_4_32"])
    np.testing.assert_array_equal(gen.pqmf.synthesis_weights.numpy(), golden["pqmf/synthesis_4_32"])
    assert gen.pqmf._cutoff_ratio == float(golden["pqmf/cutoffs"][0])
    # weight-norm registration sets g = ||v|| (w == v at init); ConvTranspose dim 0 is C_in
    ct = gen.decoder_blocks[0].conv_trans.parametrizations["weight"]
    assert ct.original0.shape == (256, 1, 1) and ct.original1.shape == (256, 128, 16)
    assert torch.allclose(ct.original0.flatten(), ct.original1.reshape(256, -1).norm(dim=1))
    for l_in, l_out in zip(golden["cut/in"], golden["cut/out"]):
        assert gen.cut_to_valid_length(torch.zeros(1, 1, int(l_in))).shape[2] == int(l_out)


def test_generator_checkpoint_round_trip_offline(tmp_path):
    """the PyTorchModelHubMixin surface of the reference generator
    (eben_generator.py:72-85, scripts/upload_eben_to_hub.py) -- save_pretrained / from_pretrained on a local
    directory: config.json carries {m, n, p}, the safetensors file the reference's state_dict keys."""
    import json

    from vibravox_amd.torch_modules.dnn.eben_generator import EBENGenerator

    torch.manual_seed(7)
    gen = EBENGenerator(m=4, n=32, p=1)
    if not hasattr(gen, "save_pretrained"):
        pytest.skip("huggingface_hub not installed")
    gen.save_pretrained(str(tmp_path))
    cfg = json.load(open(tmp_path / "config.json"))
    assert {k: cfg[k] for k in ("m", "n", "p")} == {"m": 4, "n": 32, "p": 1}
    assert (tmp_path / "model.safetensors").exists()
    back = EBENGenerator.from_pretrained(str(tmp_path))
    a, b = gen.state_dict(), back.state_dict()
    assert list(a) == list(b) and back.first_conv.weight.shape == (32, 1, 3)
    for k in a:
        assert torch.equal(a[k], b[k]), k


def test_reference_constructor_asserts():
    from vibravox_amd.lightning_modules.eben import EBENLightningModule
    from vibravox_amd.torch_modules.dnn.eben_discriminator import DiscriminatorEBEN
    from vibravox_amd.torch_modules.dsp.pqmf import PseudoQMFBanks

    with pytest.raises(AssertionError):
        PseudoQMFBanks(decimation=4, kernel_size=30)  # pqmf.py:42
    with pytest.raises(AssertionError):
        DiscriminatorEBEN(q=5, min_channels=24)  # eben_discriminator.py:64
    gen, disc = torch.nn.Linear(1, 1), torch.nn.Linear(1, 1)
    opt = lambda params: torch.optim.SGD(params, lr=0.1)
    with pytest.raises(AssertionError):
        EBENLightningModule(16000, gen, disc, opt, opt, dynamic_loss_balancing="bogus")  # eben.py:67-71
    with pytest.raises(AssertionError):
        EBENLightningModule(16000, gen, disc, opt, opt, update_discriminator_ratio=1.5)  # eben.py:76
    mod = EBENLightningModule(16000, gen, disc, opt, opt, dynamic_loss_balancing="ema")
    assert mod.automatic_optimization is False and len(mod.configure_optimizers()) == 2
    with pytest.raises(ValueError):
        PseudoQMFBanks(4, 32).forward(torch.zeros(1, 1, 64), "bogus")  # pqmf.py:215


def test_no_cpu_fallback():
    """The product path must fail loudly instead of computing on the CPU."""
    from vibravox_amd import _lib
    from vibravox_amd.torch_modules.dnn.eben_generator import EBENGenerator
    from vibravox_amd.torch_modules.losses.hinge_loss import HingeLossForDiscriminatorMelganMultiScales

    gen = EBENGenerator(4, 32, 2)
    with pytest.raises(_lib.EbenError, match="no CPU path"):
        gen(torch.zeros(1, 1, 224))
    with pytest.raises(_lib.EbenError):
        HingeLossForDiscriminatorMelganMultiScales()(embeddings=[[torch.zeros(1, 1, 8)]], target=1)
    src = "".join(open(os.path.join(dp, f)).read() for dp, _, fs in os.walk(os.path.join(ROOT, "vibravox_amd")) for f in fs if f.endswith(".py"))
    assert "import oracle" not in src and "from oracle" not in src  # the oracle is test infrastructure only


def test_conv_spec_lengths_match_torch():
    from vibravox_amd.ops import ConvSpec

    for k, s, d, p, l in [(41, 4, 1, 20, 31968), (7, 2, 3, 3, 8002), (16, 8, 1, 7, 1000), (3, 1, 9, 9, 125), (15, 1, 1, 0, 31982)]:
        ref = torch.nn.functional.conv1d(torch.zeros(1, 1, l + 2 * p), torch.zeros(1, 1, k), stride=s, dilation=d).shape[2]
        assert ConvSpec(1, 1, k, stride=s, dilation=d, pad_l=p, pad_r=p).out_len(l) == ref
    for k, s, p, l in [(16, 8, 4, 125), (4, 2, 1, 4000), (32, 4, 31, 8000)]:
        ref = torch.nn.functional.conv_transpose1d(torch.zeros(1, 1, l), torch.zeros(1, 1, k), stride=s, padding=p).shape[2]
        assert ConvSpec(1, 1, k, stride=s, pad_l=p, transposed=True).out_len(l) == ref


@pytest.fixture(scope="module")
def cref():
    out = os.path.join(ROOT, "oracle", "_build")
    os.makedirs(out, exist_ok=True)
    so = os.path.join(out, "libconv_ref.so")
    subprocess.run(["gcc", "-O2", "-shared", "-fPIC", "-o", so, os.path.join(ROOT, "oracle", "conv_ref.c")], check=True)
    return ctypes.CDLL(so)


def _dp(t):
    return t.ctypes.data_as(ctypes.POINTER(ctypes.c_double))


@pytest.mark.parametrize("case", [
    dict(c_in=8, c_out=12, k=7, stride=2, dil=3, groups=4, pad=3, reflect=False),
    dict(c_in=6, c_out=6, k=3, stride=1, dil=9, groups=1, pad=9, reflect=True),
    dict(c_in=4, c_out=8, k=8, stride=4, dil=1, groups=1, pad=3, reflect=True),
    dict(c_in=1, c_out=4, k=32, stride=4, dil=1, groups=1, pad=31, reflect=False),
])
def test_oracle_conv_against_plain_c(cref, case):
    b, l = 2, 67
    x = formula_tensor("c/x", (b, case["c_in"], l)).double()
    w = formula_tensor("c/w", (case["c_out"], case["c_in"] // case["groups"], case["k"])).double()
    bias = formula_tensor("c/b", (case["c_out"],)).double()
    ref = O.conv_layer(x, w, None, bias, stride=case["stride"], dilation=case["dil"], groups=case["groups"],
                       pad_l=case["pad"], pad_r=case["pad"], reflect=case["reflect"])
    y = np.zeros(tuple(ref.shape))
    xn, wn, bn = x.numpy().copy(), w.numpy().copy(), bias.numpy().copy()
    cref.ref_conv1d(_dp(xn), _dp(wn), _dp(bn), _dp(y), b, case["c_in"], case["c_out"], l, ref.shape[2], case["k"], case["stride"],
                    case["dil"], case["groups"], case["pad"], int(case["reflect"]))
    np.testing.assert_allclose(ref.numpy(), y, rtol=1e-12, atol=1e-12)


@pytest.mark.parametrize("k,stride,pad,groups", [(16, 8, 4, 1), (4, 2, 1, 1), (32, 4, 31, 4)])
def test_oracle_conv_transpose_against_plain_c(cref, k, stride, pad, groups):
    b, c_in, c_out, l = 2, 4, 4 if groups == 4 else 6, 23
    x = formula_tensor("ct/x", (b, c_in, l)).double()
    w = formula_tensor("ct/w", (c_in, c_out // groups, k)).double()
    op = stride - 2 if groups == 4 else 0  # the PQMF synthesis uses output_padding = M - 2
    ref = torch.nn.functional.conv_transpose1d(x, w, None, stride=stride, padding=pad, output_padding=op, groups=groups)
    y = np.zeros(tuple(ref.shape))
    xn, wn = x.numpy().copy(), w.numpy().copy()
    cref.ref_conv_transpose1d(_dp(xn), _dp(wn), _dp(y), b, c_in, c_out, l, ref.shape[2], k, stride, 1, groups, pad)
    np.testing.assert_allclose(ref.numpy(), y, rtol=1e-12, atol=1e-12)


def test_run_py_config_composition_and_overrides():
    """run.py resolves the reference's Hydra surface for the EBEN path: group selection, dotted
    overrides (a.b=c, +a.b=c, ++a.b=c), ${...} interpolation, _target_/_partial_ instantiation."""
    import run

    cfg = run.compose(["lightning_module=eben", "lightning_module.generator.p=1", "++trainer.max_steps=3", "+trainer.new_key=7"])
    assert cfg["lightning_module"]["generator"] == {"_target_": "vibravox_amd.torch_modules.dnn.eben_generator.EBENGenerator", "m": 4, "p": 1, "n": 32}
    assert cfg["lightning_module"]["discriminator"]["q"] == 4 and cfg["lightning_module"]["discriminator"]["min_channels"] == 24
    assert cfg["trainer"]["max_steps"] == 3 and cfg["trainer"]["new_key"] == 7
    assert cfg["lightning_module"]["sample_rate"] == 16000  # ${sample_rate}
    assert cfg["lightning_module"]["reconstructive_loss_freq_fn"]["fft_sizes"] == [512, 1024, 2048]
    assert cfg["lightning_module"]["dynamic_loss_balancing"] == "ema" and cfg["lightning_module"]["beta_ema"] == 0.9
    opt = cfg["lightning_module"]["generator_optimizer"]
    assert opt["lr"] == 3e-4 and opt["betas"] == [0.5, 0.9] and opt["_partial_"] is True
    with pytest.raises(KeyError):
        run.compose(["lightning_module.no_such_key=1"])
    gen = run.instantiate(cfg["lightning_module"]["generator"])
    assert type(gen).__name__ == "EBENGenerator" and gen.p == 1
    part = run.instantiate(opt)
    o = part(params=gen.parameters())
    assert type(o).__name__ == "FusedAdam" and o.defaults["lr"] == 3e-4 and o.defaults["betas"] == (0.5, 0.9)
    module = run.instantiate(cfg["lightning_module"])
    assert type(module).__name__ == "EBENLightningModule" and module.dynamic_loss_balancing == "ema"
    assert type(module.reconstructive_loss_freq_fn).__name__ == "MultiResolutionSTFTLoss"


def test_trainer_precision_selects_the_arithmetic_plan():
    """vibravox configs/trainer/ddp.yaml:23-25: the reference's knob for the step's arithmetic is ``trainer.precision``.  run.py hands it
    to ``EBENLightningModule.set_precision``: ``bf16-mixed`` is the plan bench.py measures (BASELINE config 2), the default stays the
    reference's fp32."""
    import run
    from vibravox_amd.lightning_modules.eben import DISC_MATH_PLANS, EBENLightningModule

    cfg = run.compose(["lightning_module=eben"])
    assert cfg["trainer"]["precision"] == "32-true"
    module = run.instantiate(cfg["lightning_module"]).set_precision(cfg["trainer"]["precision"])
    assert (module.disc_math, module.gen_backward_math, module.stft_math) == ("f32", "f32", None)
    cfg = run.compose(["lightning_module=eben", "trainer.precision=bf16-mixed"])
    module = run.instantiate(cfg["lightning_module"]).set_precision(cfg["trainer"]["precision"])
    assert (module.disc_math, module.gen_backward_math, module.stft_math) == ("bf16_bl", "bf16", "folded_x3")
    assert module.disc_math in DISC_MATH_PLANS and DISC_MATH_PLANS[module.disc_math]["layout"] == "bl"
    for alias, plan in (("bf16", "bf16-mixed"), (32, "32-true"), ("32-split", "32-split")):
        assert module.set_precision(alias).precision == plan
    assert module.disc_math == "bf16x6" and module.gen_backward_math == "f32" and module.stft_math == "folded_x6"
    with pytest.raises(ValueError):
        module.set_precision("fp8")
    with pytest.raises(ValueError):
        module.set_precision("16-mixed")      # Lightning's fp16 AMP: not silently mapped onto the bf16 plan
    assert all(p[0] in DISC_MATH_PLANS for p in EBENLightningModule.PRECISION_PLANS.values())


def test_bundle_layout_engine_falls_back_for_heads_it_has_no_kernel_for():
    """DiscriminatorEBENMultiScales at the reference's class default q = 3 (eben_discriminator.py:18: heads 3 -> 24) under the benchmarked
    plan: the bundle-layout engine's chain-edge kernels are built for the configured heads, so the engine of the same arithmetic plan on
    fp32 tensors at rest is constructed instead (with a warning) -- not an EBEN_EUNSUPPORTED on the first training forward."""
    from vibravox_amd.disc_engine import DiscriminatorEngine
    from vibravox_amd.disc_engine_bl import DiscriminatorEngineBL
    from vibravox_amd.lightning_modules.eben import DISC_MATH_PLANS
    from vibravox_amd.torch_modules.dnn.eben_discriminator import DiscriminatorEBENMultiScales

    plan = DISC_MATH_PLANS["bf16_bl"]
    d4, d3 = DiscriminatorEBENMultiScales(q=4, min_channels=24), DiscriminatorEBENMultiScales(q=3, min_channels=24)
    assert DiscriminatorEngineBL.unsupported(d4) is None and type(DiscriminatorEngine(d4, plan)) is DiscriminatorEngineBL
    assert "3 -> 24" in DiscriminatorEngineBL.unsupported(d3)
    with pytest.warns(UserWarning, match="fp32 tensors at rest"):
        eng = DiscriminatorEngine(d3, plan)
    assert type(eng) is DiscriminatorEngine and eng.math is plan   # the step does not rebuild it every call
