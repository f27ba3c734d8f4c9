EBEN_DISC_MATH=bf16 timeout 600 python tools/cpu_time.py 2>&1 | tail -2
EBEN_DISC_MATH=bf16 timeout 600 python -c "
import sys, time, torch, cProfile, pstats
sys.path.insert(0,'.')
import bench
dev = torch.device('cuda', 0)
mod = bench.build_module(dev, 1234)
batch = bench.synthetic_batch(32, 32000, 1234, dev)
for _ in range(3): mod.training_step(batch)
torch.cuda.synchronize()
pr = cProfile.Profile(); pr.enable()
for _ in range(5): mod.training_step(batch)
pr.disable(); torch.cuda.synchronize()
pstats.Stats(pr).sort_stats('tottime').print_stats(22)
" 2>&1 | tail -40
