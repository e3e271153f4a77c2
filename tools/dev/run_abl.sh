mkdir -p gpurun_out/r03
DSIM_LIB=$PWD/disimpy_amd/csrc/libdsim_hip_ablate.so timeout -k 5 300 python tools/ablate_short.py c5b 1000000 > gpurun_out/r03/ablate_short_1m.txt 2>&1
DSIM_LIB=$PWD/disimpy_amd/csrc/libdsim_hip_ablate.so timeout -k 5 300 python tools/ablate_short.py c5b 8000000 > gpurun_out/r03/ablate_short_8m.txt 2>&1
cat gpurun_out/r03/ablate_short_1m.txt gpurun_out/r03/ablate_short_8m.txt | tail -12
