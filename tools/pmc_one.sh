#!/bin/bash
# usage: pmc_one.sh tag W H counters...
tag=$1; W=$2; H=$3; shift 3
cd /tmp && export TMPDIR=/tmp
R="$GRAFT_REPO_ROOT"; O="$R/gpurun_out"
timeout 600 rocprofv3 --pmc "$@" -d $O/${tag} -o p -- python3 $R/tools/experiments/frames_only.py $W $H 60 chain 1 > /dev/null 2> $O/${tag}.log
cd "$R"; python3 tools/pmc_dump.py gpurun_out/${tag} k_ | grep -E "warp_bin|unsharp|collapse_level<true>"
rm -rf gpurun_out/${tag}
