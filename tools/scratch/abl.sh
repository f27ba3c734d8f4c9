mkdir -p gpurun_out/r3abl
python tools/layer_bench_bl.py --iters 20 > gpurun_out/r3abl/main.txt 2>&1
for v in t1 t4 t8 t32 t64 t128 t256 t5; do EBEN_HIP_LIB=$PWD/variants/$v/libeben_hip.so timeout 300 python tools/layer_bench_bl.py --iters 20 > gpurun_out/r3abl/$v.txt 2>&1; done
python tools/layer_bench_bl.py --iters 20 > gpurun_out/r3abl/main2.txt 2>&1
tail -1 gpurun_out/r3abl/*.txt
