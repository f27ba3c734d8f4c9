"""Where the bf16 discriminator-gradient error comes from: the same state / batch through the engine step with the three
contraction families (forward, input gradient, weight gradient) of the discriminators switched between exact fp32 and
bf16 operands one at a time, and per sub-discriminator layer.  Prints the whole-vector and per-tensor relative L2 distance
of the discriminator gradient (Adam's first moment after one step = (1 - beta1) * grad) to the all-fp32 step, and the
cancellation ratio R = (|G_fake| + |G_real|) / |G_fake + G_real| of the two hinge branches (autograd path, fp32)."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import bench
from vibravox_amd import ops
dev = torch.device("cuda", 0)
B, T = int(os.environ.get("B", "8")), int(os.environ.get("T", "16000"))
F, H, X = ops.MATH_F32, ops.MATH_BF16, ops.MATH_BF16X2
DATA = os.environ.get("DATA", "noise")


def make():
    mod = bench.build_module(dev, 1234)
    if DATA == "formula":
        from formula import formula_audio
        batch = {"audio_body_conducted": formula_audio("bf/bc", B, T).to(dev), "audio_airborne": formula_audio("bf/air", B, T).to(dev)}
    else:
        batch = bench.synthetic_batch(B, T, 1234, dev)
    return mod, batch


def run(plan, seed_weights=None):
    mod, batch = make()
    mod.disc_math = plan
    if seed_weights is not None:
        from vibravox_amd.disc_engine import DiscriminatorEngine
        from vibravox_amd.lightning_modules.eben import DISC_MATH_PLANS
        mod._disc_engine = DiscriminatorEngine(mod.discriminator, DISC_MATH_PLANS[plan] if isinstance(plan, str) else plan)
        mod._disc_engine.seed_weights = seed_weights
    mod.training_step(batch)
    torch.cuda.synchronize()
    opt = mod._optimizers[1]
    names = {id(p): n for n, p in mod.discriminator.named_parameters()}
    return {names[id(p)]: opt.state[p]["exp_avg"].double().cpu() for grp in opt.param_groups for p in grp["params"] if "exp_avg" in opt.state.get(p, {})}


def dist(a, b, sel=lambda n: True):
    num = sum(float((a[n] - b[n]).norm() ** 2) for n in a if sel(n)) ** 0.5
    den = sum(float(a[n].norm() ** 2) for n in a if sel(n)) ** 0.5
    return num / (den + 1e-300)


def branches():
    """fp32 autograd path: gradients of fake_loss and real_loss separately."""
    mod, batch = make()
    gen, disc = mod.generator, mod.discriminator
    x = gen.cut_to_valid_length(batch["audio_body_conducted"]); y = gen.cut_to_valid_length(batch["audio_airborne"])
    with torch.no_grad():
        enh, bands = gen(x)
        bref = gen.pqmf.forward(y, "analysis")
    out = {}
    for name, (b_, a_, tgt) in {"fake": (bands, enh, -1), "real": (bref, y, 1)}.items():
        disc.zero_grad()
        mod.adversarial_loss_fn(embeddings=disc(bands=b_, audio=a_), target=tgt).backward()
        out[name] = {n: p.grad.double().cpu().clone() for n, p in disc.named_parameters() if p.grad is not None}
    return out


ref = run("f32")
br = branches()
tot = {n: br["fake"][n] + br["real"][n] for n in br["fake"]}
print(f"# B={B} T={T} data={DATA}")
print(f"engine f32 vs autograd f32 (fake+real): {dist({n: 0.5 * v for n, v in tot.items()}, ref):.3e}")
nf = sum(float(v.norm() ** 2) for v in br["fake"].values()) ** 0.5
nr = sum(float(v.norm() ** 2) for v in br["real"].values()) ** 0.5
nt = sum(float(v.norm() ** 2) for v in tot.values()) ** 0.5
print(f"cancellation: |G_fake| {nf:.3e} |G_real| {nr:.3e} |sum| {nt:.3e}  R = {(nf + nr) / nt:.1f}")
plans = {"bf16 all": (H, H, H), "fwd only": (H, F, F), "dx+dw": (F, H, H),
         "pq f32fwd": {"pqmf": (F, H, H), "melgan": H}, "pq f32fwd x2dw": {"pqmf": (F, H, X), "melgan": H}, "pq f32": {"pqmf": F, "melgan": H},
         "pq f32, mel x2dw": {"pqmf": F, "melgan": (H, H, X)}, "mel fwd only": {"pqmf": F, "melgan": (H, F, F)}}
if os.environ.get("PLANS"):
    plans = {k: v for k, v in plans.items() if k in os.environ["PLANS"].split(",") or k == "bf16 all"}
res = {}
for name, plan in plans.items():
    res[name] = run(plan)
    print(f"{name:10s} whole {dist(ref, res[name]):.3e}   pqmf0 {dist(ref, res[name], lambda n: n.startswith('pqmf_discriminators.0')):.3e}"
          f"  melgan {dist(ref, res[name], lambda n: n.startswith('melgan')):.3e}")
print("per tensor (bf16 all):  R = (|fake|+|real|)/|sum|,  rel-L2 vs fp32")
for n in ref:
    r = float((br["fake"][n].norm() + br["real"][n].norm()) / (tot[n].norm() + 1e-300))
    print(f"  {n:75s} R {r:8.1f}  " + "  ".join(f"{k} {dist(ref, v, lambda q: q == n):.2e}" for k, v in res.items()))

# each hinge branch on its own (no cancellation): fp32 vs the step's bf16 plan
for name, w in (("fake", (1.0, 1.0, 0.0)), ("real", (1.0, 0.0, 1.0))):
    a, b = run("f32", w), run("bf16", w)
    print(f"branch {name}: engine bf16 vs f32 rel-L2 {dist(a, b):.3e}   (f32 engine vs autograd: {dist({n: 0.5 * v for n, v in br[name].items()}, a):.3e})")
tot_b = run("bf16")
num = sum(float((ref[n] - tot_b[n]).norm() ** 2) for n in ref) ** 0.5
print(f"'bf16' plan: |dG| / |G| = {dist(ref, tot_b):.3e};  |dG| / (|G_fake| + |G_real|) = {num / (0.5 * (nf + nr)):.3e}")
