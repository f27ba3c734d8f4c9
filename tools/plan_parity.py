"""One step at BASELINE config 2 from the same state in several discriminator math plans: relative L2 of the generator /
discriminator gradient against the fp32 step, and the step time of each plan (20 timed steps).  Usage: python tools/plan_parity.py [plans...]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench

DEV = torch.device("cuda")
args = [a for a in sys.argv[1:] if not a.startswith("--stft=")]
stft = ([a.split("=", 1)[1] for a in sys.argv[1:] if a.startswith("--stft=")] or [None])[0]   # MRSTFT math of the plans (reference: exact fp32)
plans = args or ["bf16", "bf16_x6fwd", "bf16_f32fwd", "bf16x6", "bf16_plain"]


def one_step(plan, gen_bwd, stft_math="folded"):
    mod = bench.build_module(DEV, 1234)
    batch = bench.synthetic_batch(32, 32000, 1234, DEV)
    mod.disc_math, mod.gen_backward_math, mod.stft_math = plan, gen_bwd, stft_math
    mod.training_step(batch)
    torch.cuda.synchronize()
    m = []
    for opt in mod.optimizers():
        m.append(torch.cat([opt.state[p]["exp_avg"].double().flatten().cpu() for grp in opt.param_groups for p in grp["params"] if "exp_avg" in opt.state.get(p, {})]))
    # timing
    for _ in range(3): mod.training_step(batch)
    import gc; gc.collect()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20): mod.training_step(batch)
    torch.cuda.synchronize()
    return m, (time.perf_counter() - t0) / 20 * 1e3


ref, t32 = one_step("f32", "f32")
print(f"{'f32':12s} {t32:7.2f} ms/step")
for plan in plans:
    gb = "f32" if plan in ("f32", "bf16x6") else "bf16"
    m, t = one_step(plan, gb, stft or ("folded" if plan == "f32" else "folded_x6"))
    rel = [float((a - b).norm() / a.norm()) for a, b in zip(ref, m)]
    print(f"{plan:12s} (MRSTFT {stft or 'folded_x6'}) {t:7.2f} ms/step   generator grad {rel[0]:.3e}   discriminator grad {rel[1]:.3e}   (generator backward {gb})")
