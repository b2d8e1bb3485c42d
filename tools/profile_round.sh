#!/bin/bash
# Collects the round's rocprofv3 evidence on the GPU box: kernel trace + stats of the bench command, and the two
# HBM-traffic counter passes (FETCH_SIZE and WRITE_SIZE do not fit one pass on gfx950).  Usage:
#   gpurun -- bash tools/profile_round.sh <tag>         ->  gpurun_out/<tag>_{trace,fetch,write}[_4k]/
tag=${1:-r01_l}
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
B="python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --headline-only"
K4="--width 3840 --height 2160 --frames 20"
timeout 600 rocprofv3 --kernel-trace --stats -d gpurun_out/${tag}_trace -o t -- $B > gpurun_out/${tag}_trace.json 2> gpurun_out/${tag}_trace.log
timeout 600 rocprofv3 --pmc FETCH_SIZE -d gpurun_out/${tag}_fetch -o f -- $B > /dev/null 2> gpurun_out/${tag}_fetch.log
timeout 600 rocprofv3 --pmc WRITE_SIZE -d gpurun_out/${tag}_write -o w -- $B > /dev/null 2> gpurun_out/${tag}_write.log
timeout 600 rocprofv3 --kernel-trace --stats -d gpurun_out/${tag}_trace_4k -o t -- $B $K4 > gpurun_out/${tag}_trace_4k.json 2> gpurun_out/${tag}_trace_4k.log
timeout 600 rocprofv3 --pmc FETCH_SIZE -d gpurun_out/${tag}_fetch_4k -o f -- $B $K4 > /dev/null 2> gpurun_out/${tag}_fetch_4k.log
timeout 600 rocprofv3 --pmc WRITE_SIZE -d gpurun_out/${tag}_write_4k -o w -- $B $K4 > /dev/null 2> gpurun_out/${tag}_write_4k.log
# pair set-up (poppy_hip_pair_begin from raw BGR), three calls: kernel trace of the filter chain, ORB and matcher launches
timeout 600 rocprofv3 --kernel-trace --stats -d gpurun_out/${tag}_setup_trace -o t -- python3 tools/experiments/pair_begin_time.py > gpurun_out/${tag}_setup.log 2>&1
python3 bench.py --steps 10 --warmup 3 > gpurun_out/${tag}_bench.json
python3 bench.py --steps 5 --warmup 2 $K4 --no-cpu-baseline > gpurun_out/${tag}_bench_4k.json
ls -la gpurun_out/${tag}_*
