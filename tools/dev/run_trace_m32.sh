mkdir -p gpurun_out/r03
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout -k 5 500 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/r03/trace_m32 -- python3 bench.py --workload c5b --walkers 10000000 --measurements 32 --steps 1 --warmup 0 --no-cpu-baseline --no-secondary --no-signal-check > gpurun_out/r03/trace_m32.json 2> gpurun_out/r03/trace_m32.err
python tools/timeline.py gpurun_out/r03/trace_m32 > gpurun_out/r03/timeline_m32_recorded.txt 2>&1
rm -rf gpurun_out/r03/trace_m32
head -24 gpurun_out/r03/timeline_m32_recorded.txt; tail -14 gpurun_out/r03/timeline_m32_recorded.txt
