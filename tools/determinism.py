"""Run the same two train steps on two identically initialised modules and compare every parameter bit for bit
(a difference means a race between streams: every kernel on the path is order-deterministic)."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden"))
import bench
dev = torch.device("cuda", 0)
def run():
    mod = bench.build_module(dev, 1234)
    batch = bench.synthetic_batch(int(os.environ.get("B", "8")), 16000, 1234, dev)
    for _ in range(int(os.environ.get("STEPS", "3"))):
        mod.training_step(batch)
    torch.cuda.synchronize()
    out = {k: v.clone() for k, v in list(mod.generator.state_dict().items()) + [("D." + k, v) for k, v in mod.discriminator.state_dict().items()]}
    for oi, opt in enumerate(mod._optimizers):
        for gi, grp in enumerate(opt.param_groups):
            for pi, p in enumerate(grp["params"]):
                if "exp_avg" in opt.state.get(p, {}):
                    out[f"opt{oi}.m.{pi}.{tuple(p.shape)}"] = opt.state[p]["exp_avg"].clone()
    return out
a = run(); b = run()
bad = [(k, float((a[k] - b[k]).abs().max())) for k in a if not torch.equal(a[k], b[k])]
print("differing tensors:", len(bad), "of", len(a)); print(bad[:8])
