mkdir -p gpurun_out/r03
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout -k 5 500 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/r03/trace1250k -- python3 bench.py --workload c5b --walkers 1250000 --steps 1 --warmup 1 --no-cpu-baseline --no-secondary --no-signal-check > gpurun_out/r03/trace1250k.json 2> gpurun_out/r03/trace1250k.err
python tools/timeline.py gpurun_out/r03/trace1250k > gpurun_out/r03/timeline_1250k.txt 2>&1
rm -rf gpurun_out/r03/trace1250k
cat gpurun_out/r03/timeline_1250k.txt | tail -80
