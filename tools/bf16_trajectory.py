"""Do the discriminator's arithmetic modes train the same way?  N consecutive train steps from the same initial state on
the same sequence of batches, once per mode; prints every logged loss per step and the relative deviation from the fp32
run (plus the relative L2 distance of the parameters at the end, in units of the distance travelled)."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
dev = torch.device("cuda", 0)
B, T, N = int(os.environ.get("B", "4")), int(os.environ.get("T", "16000")), int(os.environ.get("N", "40"))
MODES = os.environ.get("MODES", "f32,bf16,bf16_plain").split(",")


def run(mode):
    # "f32+p": the fp32 step on inputs perturbed by one part in 2^23 (one fp32 ulp): how fast does this GAN's training amplify
    # the smallest possible difference?  That divergence is the yardstick for the other modes'.
    perturb = mode.endswith("+p")
    mode = mode.split("+")[0]
    mod = bench.build_module(dev, 1234)
    mod.disc_math = mode
    mod.gen_backward_math = "f32" if mode == "f32" else "bf16"
    p0 = torch.cat([p.detach().flatten().double().cpu() for p in mod.discriminator.parameters()])
    g0 = torch.cat([p.detach().flatten().double().cpu() for p in mod.generator.parameters()])
    torch.manual_seed(7)   # the step's own draw (update_discriminator_ratio)
    logs = []
    for i in range(N):
        batch = bench.synthetic_batch(B, T, 1000 + i, dev)
        if perturb:
            g = torch.Generator(device="cpu").manual_seed(5000 + i)
            batch = {k: v * (1 + 2.0 ** -23 * (2 * torch.rand(v.shape, generator=g).to(dev) - 1)) for k, v in batch.items()}
        mod.training_step(batch)
        logs.append({k: float(v) for k, v in mod.logged.items()})
    torch.cuda.synchronize()
    p1 = torch.cat([p.detach().flatten().double().cpu() for p in mod.discriminator.parameters()])
    g1 = torch.cat([p.detach().flatten().double().cpu() for p in mod.generator.parameters()])
    return logs, (p0, p1), (g0, g1)


res = {m: run(m) for m in MODES}
ref = res[MODES[0]]
keys = list(ref[0][0].keys())
print(f"# B={B} T={T} steps={N} modes={MODES}")
for m in MODES[1:]:
    worst = {k: 0.0 for k in keys}
    for i in range(N):
        for k in keys:
            a, b = ref[0][i][k], res[m][0][i][k]
            worst[k] = max(worst[k], abs(a - b) / (abs(a) + 1e-12))
    print(f"{m}: worst relative deviation of the logged values over {N} steps:")
    for k in keys:
        print(f"   {k:48s} {worst[k]:.3e}   (last: {ref[0][-1][k]:+.5e} vs {res[m][0][-1][k]:+.5e})")
    for name, idx in (("discriminator", 1), ("generator", 2)):
        a0, a1 = ref[idx]
        b0, b1 = res[m][idx]
        print(f"   {name} parameters: |theta_m - theta_f32| / |theta_f32 - theta_0| = {float((b1 - a1).norm() / (a1 - a0).norm()):.3e}")
for i in range(N):
    print(f"step {i:3d} " + "  ".join(f"{m}: D real {res[m][0][i]['train/discriminator/real_loss']:.5f} fake {res[m][0][i]['train/discriminator/fake_loss']:.5f}"
                                    f" fm {res[m][0][i]['train/generator/feature_matching_loss']:.5f}" for m in MODES))
