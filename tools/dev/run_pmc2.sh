mkdir -p gpurun_out/r03
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
cp profiles/kernel_counters.json /tmp/kc_backup.json
timeout -k 10 900 python3 tools/pmc_profile.py c5b --walkers 10000000 --tag r03tmp10m --passes sq,ta > gpurun_out/r03/pmc_c5b_10m_ta.txt 2>&1
timeout -k 10 900 python3 tools/pmc_profile.py c5b --walkers 1250000 --tag r03tmp1m --passes sq,ta > gpurun_out/r03/pmc_c5b_1m_ta.txt 2>&1
cp profiles/r03tmp*_c5b_pmc.json gpurun_out/r03/ 2>/dev/null
tail -22 gpurun_out/r03/pmc_c5b_10m_ta.txt; tail -22 gpurun_out/r03/pmc_c5b_1m_ta.txt
