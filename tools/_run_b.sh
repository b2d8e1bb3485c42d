cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
python tools/_pbtime.py
rocprofv3 --kernel-trace --stats -d gpurun_out/prof_pb -o pb -- python3 tools/_pbtime.py > /dev/null 2>&1
