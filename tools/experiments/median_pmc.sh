# pipe counters of one median kernel: median_pmc.sh <ksize> <form> <content>
cd /tmp && export TMPDIR=/tmp
R="$GRAFT_REPO_ROOT"; O="$R/gpurun_out"
k=${1:-49}; f=${2:-3}; c=${3:-photo}
i=0
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES GRBM_GUI_ACTIVE" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA" "SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT" "SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_BRANCH SQ_INSTS_SMEM"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $grp -d $O/mp_$i -o p -- python3 $R/tools/experiments/median_one.py 1920 1080 $k $f $c 3 > /dev/null 2> $O/mp_$i.log
done
cd "$R"
for j in $(seq 1 $i); do python3 tools/pmc_dump.py gpurun_out/mp_$j k_median; rm -rf gpurun_out/mp_$j; done
