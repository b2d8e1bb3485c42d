#!/bin/bash
# Pipe-utilisation counters of the chained frame loop, one rocprofv3 --pmc pass per counter group (counter passes carry no trace
# options).  Usage: gpurun -- bash tools/pmc_pipes.sh <tag> <W> <H>   ->  gpurun_out/<tag>_pipes.md
tag=${1:-pipes}; W=${2:-1920}; H=${3:-1080}
cd /tmp && export TMPDIR=/tmp
R="$GRAFT_REPO_ROOT"; O="$R/gpurun_out"
F="python3 $R/tools/experiments/frames_only.py"
i=0
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES" "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU" "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY" "SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_ANY SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL" "TA_TA_BUSY TA_BUFFER_WAVEFRONTS TA_FLAT_WAVEFRONTS" "GRBM_GUI_ACTIVE TCC_BUSY"; do
  i=$((i+1))
  timeout 600 rocprofv3 --pmc $grp -d $O/${tag}_p$i -o p -- $F $W $H 60 chain 1 > /dev/null 2> $O/${tag}_p$i.log
done
cd "$R"
{ echo "# pipe counters, frames_only.py $W $H 60 chain 1 (averages per launch)"; for j in $(seq 1 $i); do python3 tools/pmc_dump.py gpurun_out/${tag}_p$j k_; done; } > gpurun_out/${tag}_pipes.md
for j in $(seq 1 $i); do rm -rf gpurun_out/${tag}_p$j; done
wc -l gpurun_out/${tag}_pipes.md
