mkdir -p gpurun_out/r03
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
cp profiles/kernel_counters.json /tmp/kc_backup.json
timeout -k 10 900 python3 tools/pmc_profile.py c5b --walkers 1000000 --tag r03tmp --passes sq,mem > gpurun_out/r03/pmc_c5b_1m.txt 2>&1
cp profiles/r03tmp_c5b_pmc.json gpurun_out/r03/ 2>/dev/null
cat gpurun_out/r03/pmc_c5b_1m.txt | tail -30
