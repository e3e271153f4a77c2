mkdir -p gpurun_out/r03
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
cp profiles/kernel_counters.json /tmp/kc_backup.json
timeout -k 10 280 python3 tools/pmc_profile.py c5b --walkers 1000000 --tag r03tmp1m --passes ta > gpurun_out/r03/pmc_c5b_1m_ta.txt 2>&1
tail -25 gpurun_out/r03/pmc_c5b_1m_ta.txt
