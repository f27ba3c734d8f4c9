for d in 0 1 2 4 8 3 7; do
lib=$PWD/gpurun_exp/dbg$d/libeben_hip.so; [ $d = 0 ] && lib=$PWD/vibravox_amd/lib/libeben_hip.so
echo "== DBG $d"; EBEN_HIP_LIB=$lib timeout 600 python tools/layer_bench.py --batch 64 --filter melgan_discriminator.discriminator.4 --math bf16 2>&1 | cut -c1-58,66-130 | grep "discriminator.4.0"
done
