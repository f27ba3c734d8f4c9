"""cProfile of the host side of the engine train step (30 graph-replayed steps, the GPU running behind): where the enqueue time goes.
Usage: python tools/host_profile.py [--sort tottime|cumulative] [--top N]"""
import argparse, cProfile, os, pstats, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from vibravox_amd import ops

ap = argparse.ArgumentParser(); ap.add_argument("--sort", default="tottime"); ap.add_argument("--top", type=int, default=45); a = ap.parse_args()
dev = torch.device("cuda", 0)
mod = bench.build_module(dev, 1234)
mod.disc_math = "bf16_bl"; mod.gen_backward_math = "bf16"; mod.stft_math = "folded_x3"
batch = bench.synthetic_batch(32, 32000, 1234, dev)
n = 0
while n < 16 and (n < 6 or ops.graphs_pending()):
    mod.training_step(batch); n += 1
torch.cuda.synchronize()
import gc; gc.collect(); gc.freeze()
pr = cProfile.Profile()
pr.enable()
for _ in range(30):
    mod.training_step(batch)
pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr, stream=sys.stdout)
st.strip_dirs().sort_stats(a.sort).print_stats(a.top)
if os.environ.get("CALLERS"):
    st.print_callers(os.environ["CALLERS"])
