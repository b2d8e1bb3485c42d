#!/bin/bash
# Collects the round's rocprofv3 evidence on the GPU box: kernel trace + stats of the bench command, of the chained frame loop at
# 1080p and 4K and of the pair set-up, and the two HBM-traffic counter passes (FETCH_SIZE and WRITE_SIZE do not fit one pass on
# gfx950; counter passes run with --pmc only).  Usage:  gpurun -- bash tools/profile_round.sh <tag>   ->  gpurun_out/<tag>_*/
tag=${1:-r02}
cd /tmp && export TMPDIR=/tmp
R="$GRAFT_REPO_ROOT"
B="python3 $R/bench.py --steps 4 --warmup 1 --headline-only"
F="python3 $R/tools/experiments/frames_only.py"
O="$R/gpurun_out"
timeout 600 rocprofv3 --kernel-trace --stats -d $O/${tag}_bench_trace -o t -- $B > $O/${tag}_bench_trace.json 2> $O/${tag}_bench_trace.log
for sz in "1920 1080" "3840 2160"; do set -- $sz
  timeout 600 rocprofv3 --kernel-trace --stats -d $O/${tag}_chain_$1 -o t -- $F $1 $2 60 chain 3 > $O/${tag}_chain_$1.log 2>&1
  timeout 600 rocprofv3 --pmc FETCH_SIZE -d $O/${tag}_fetch_$1 -o f -- $F $1 $2 60 chain 1 > /dev/null 2> $O/${tag}_fetch_$1.log
  timeout 600 rocprofv3 --pmc WRITE_SIZE -d $O/${tag}_write_$1 -o w -- $F $1 $2 60 chain 1 > /dev/null 2> $O/${tag}_write_$1.log
done
timeout 600 rocprofv3 --kernel-trace --stats -d $O/${tag}_setup_trace -o t -- python3 $R/tools/experiments/pair_begin_time.py > $O/${tag}_setup.log 2>&1
cd "$R"
python3 bench.py > gpurun_out/${tag}_bench.json 2> gpurun_out/${tag}_bench.err
ls gpurun_out | grep "^${tag}_"
