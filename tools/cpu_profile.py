import sys, os, time, torch, cProfile, pstats
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
dev = torch.device('cuda', 0)
mod = bench.build_module(dev, 1234)
mod.disc_math = os.environ.get("EBEN_DISC_MATH", "bf16_bl"); mod.gen_backward_math = "bf16"; mod.stft_math = "folded_x3"
batch = bench.synthetic_batch(32, 32000, 1234, dev)
for _ in range(8): mod.training_step(batch)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(10): mod.training_step(batch)
t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
print(f"CPU enqueue {1e3*(t1-t0)/10:.1f} ms/step, GPU complete {1e3*(t2-t0)/10:.1f} ms/step")
pr = cProfile.Profile(); pr.enable()
for _ in range(5): mod.training_step(batch)
pr.disable(); torch.cuda.synchronize()
pstats.Stats(pr).sort_stats('tottime').print_stats(int(os.environ.get("TOP", "18")))
