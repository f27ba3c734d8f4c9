import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vibravox_amd.torch_modules.losses.mrstft_loss import MultiResolutionSTFTLoss
dev = torch.device("cuda")
loss = MultiResolutionSTFTLoss(fft_sizes=(512, 1024, 2048), hop_sizes=(50, 120, 240), win_lengths=(240, 600, 1200), sample_rate=16000, perceptual_weighting=True).to(dev)
x = (0.1 * torch.randn(32, 1, 31968, device=dev)).requires_grad_(True); y = 0.1 * torch.randn(32, 1, 31968, device=dev)
loss.stft_math = sys.argv[1] if len(sys.argv) > 1 else loss.stft_math
def run():
    l = loss(x, y); l.backward(); x.grad = None
for _ in range(3): run()
torch.cuda.synchronize()
e = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
tf = tb = 0
for _ in range(10):
    e[0].record(); l = loss(x, y); e[1].record(); l.backward(); e[2].record(); torch.cuda.synchronize()
    tf += e[0].elapsed_time(e[1]); tb += e[1].elapsed_time(e[2]); x.grad = None
print(f"mrstft [{loss.stft_math}] fwd {tf/10:.3f} ms  bwd {tb/10:.3f} ms")
# value and gradient against the exact-fp32 folded form
if loss.stft_math != "folded":
    l = loss(x, y); l.backward(); g = x.grad.clone(); x.grad = None
    loss.stft_math = "folded"
    l0 = loss(x, y); l0.backward()
    print(f"  vs folded: loss rel {abs(float(l) - float(l0)) / abs(float(l0)):.2e}, gradient rel L2 {float((g - x.grad).norm() / x.grad.norm()):.2e}")
