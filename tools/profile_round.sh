#!/bin/bash
# Collects the round's rocprofv3 evidence on the GPU box IN ONE CALL, so that every file of a tag comes from one run on one box: kernel trace + stats of the
# bench command, of the chained frame loop at 1080p and 4K and of the pair set-up, the two HBM-traffic counter passes (FETCH_SIZE and WRITE_SIZE do not fit one
# pass on gfx950; counter passes run with --pmc only), THEN the summaries (tools/make_profile_docs.sh, on the box: profiles/<tag>_trace.md, _pmc.md,
# <round>_warp_pmc.json, <round>_warp_facts.json), THEN the default bench line — whose `roofline.from_profiles` and `roofline.traffic` therefore quote the
# traces of this same call.  POPPY_RUN_ID (hostname + UTC time) is written into every file.
# Usage:  gpurun -- bash tools/profile_round.sh <tag>   ->  gpurun_out/<tag>_*/ (raw databases) and gpurun_out/<tag>_profiles/ (copy into profiles/)
tag=${1:-r02}
export POPPY_RUN_ID="$(hostname)-$(date -u +%Y%m%dT%H%M%SZ)-$tag"
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
bash tools/make_profile_docs.sh ${tag} > gpurun_out/${tag}_docs.log 2>&1          # (on the box: bench.py below reads what this writes)
python3 bench.py > gpurun_out/${tag}_bench.json 2> gpurun_out/${tag}_bench.err
cp gpurun_out/${tag}_bench.json profiles/${tag}_bench.json
mkdir -p gpurun_out/${tag}_profiles
cp profiles/${tag}_* profiles/${tag%%_*}_warp_pmc.json profiles/${tag%%_*}_warp_facts.json gpurun_out/${tag}_profiles/ 2>/dev/null
echo "$POPPY_RUN_ID" > gpurun_out/${tag}_profiles/RUN_ID
ls gpurun_out | grep "^${tag}_"
