mkdir -p gpurun_out/r03
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 120 rocprofv3 --list-avail > gpurun_out/r03/avail.txt 2>&1
grep -o -E "(TCP|TA|TD|TCC|SQ|GRBM)_[A-Za-z0-9_]+" gpurun_out/r03/avail.txt | sort -u | tr '\n' ' ' | head -c 9000
