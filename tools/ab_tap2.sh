for f in ops disc_engine _lib; do cp vibravox_amd/$f.py /tmp/${f}_new.py; done
for rep in 1 2; do
for f in ops disc_engine _lib; do cp tools/ab_old/$f.py vibravox_amd/$f.py; done
echo "== HEAD (per-layer wn launches)"; python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-f32-leg 2>&1 | grep "GPU:"
for f in ops disc_engine _lib; do cp /tmp/${f}_new.py vibravox_amd/$f.py; done
echo "== multi-tensor wn"; python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-f32-leg 2>&1 | grep "GPU:"
done
