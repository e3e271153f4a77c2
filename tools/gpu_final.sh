# Final measurements of a round on the GPU box: the default bench line, its kernel-trace stats, the counter
# passes at the bench's own size, the phase timers and traversal statistics of the development builds.
TAG=${1:-r02}
mkdir -p gpurun_out/$TAG
python bench.py > gpurun_out/$TAG/bench_default.json 2> gpurun_out/$TAG/bench_default.err; echo "bench rc=$?"
tail -c 1500 gpurun_out/$TAG/bench_default.json
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export DSIM_BENCH_POSITIONS_CACHE=/tmp/pos_final_c5b.npy  # the profiled runs below share one device fill
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/$TAG/kstats -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-secondary --no-signal-check > gpurun_out/$TAG/kstats_bench.json 2> gpurun_out/$TAG/kstats.err
find gpurun_out/$TAG/kstats -name "*kernel_stats.csv" -exec cp {} gpurun_out/$TAG/${TAG}_c5b_bench_kernel_stats.csv \;
head -5 gpurun_out/$TAG/${TAG}_c5b_bench_kernel_stats.csv
rm -rf gpurun_out/$TAG/kstats
python3 tools/pmc_profile.py c5b --walkers 10000000 --tag $TAG > gpurun_out/$TAG/pmc_c5b.txt 2>&1; cat gpurun_out/$TAG/pmc_c5b.txt
cp profiles/${TAG}_c5b_pmc.json profiles/kernel_counters.json gpurun_out/$TAG/
unset DSIM_BENCH_POSITIONS_CACHE
DSIM_LIB=disimpy_amd/csrc/libdsim_hip_timers.so python tools/mesh_timers.py c5b > gpurun_out/$TAG/${TAG}_c5b_phase_timers.txt 2>&1
DSIM_LIB=disimpy_amd/csrc/libdsim_hip_stats.so python tools/mesh_stats.py c5b > gpurun_out/$TAG/${TAG}_c5b_traversal_stats.txt 2>&1
cat gpurun_out/$TAG/${TAG}_c5b_phase_timers.txt gpurun_out/$TAG/${TAG}_c5b_traversal_stats.txt
