"""What bf16 discriminator contractions cost the train step: same weights / batch through the fp32 and the bf16
engine; relative differences of the logged losses, balancing norms / lambdas, and (through Adam's first moment
after one step = (1-beta1) * grad) the relative L2 distance of every parameter gradient."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
dev = torch.device("cuda", 0)
B, T = int(os.environ.get("B", "8")), int(os.environ.get("T", "16000"))

def run(math):
    mod = bench.build_module(dev, 1234)
    mod.disc_math = math
    batch = bench.synthetic_batch(B, T, 1234, dev)
    mod.training_step(batch)
    torch.cuda.synchronize()
    logs = {k: float(v) for k, v in mod.logged.items()}
    norms = torch.stack(mod.last_norms).cpu(); lam = torch.stack(mod.last_lambdas).cpu()
    m = []
    for oi, opt in enumerate(mod._optimizers):
        for grp in opt.param_groups:
            for p in grp["params"]:
                if "exp_avg" in opt.state.get(p, {}): m.append((oi, tuple(p.shape), opt.state[p]["exp_avg"].double().cpu()))
    return logs, norms, lam, m

a, b = run("f32"), run("bf16")
for k in a[0]:
    print(f"{k:48s} f32 {a[0][k]:+.6e}  bf16 {b[0][k]:+.6e}  rel {abs(a[0][k]-b[0][k])/(abs(a[0][k])+1e-30):.2e}")
print("norms  rel", ((a[1]-b[1]).abs()/a[1].abs()).tolist())
print("lambda rel", ((a[2]-b[2]).abs()/a[2].abs()).tolist())
for oi, name in ((0, "generator"), (1, "discriminator")):
    rels = [float((x[2]-y[2]).norm()/(x[2].norm()+1e-30)) for x, y in zip(a[3], b[3]) if x[0] == oi]
    num = sum(float((x[2]-y[2]).norm()**2) for x, y in zip(a[3], b[3]) if x[0] == oi) ** 0.5
    den = sum(float(x[2].norm()**2) for x in a[3] if x[0] == oi) ** 0.5
    rels_s = sorted(rels)
    print(f"{name}: grad rel-L2 whole {num/den:.3e}  per-tensor median {rels_s[len(rels_s)//2]:.3e}  max {rels_s[-1]:.3e}")
