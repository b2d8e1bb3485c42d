# pipe counters of k_tile_expand / k_warp_bin in the 4K (or WxH) frame loop: tile_expand_pmc.sh [W H]
cd /tmp && export TMPDIR=/tmp
R="$GRAFT_REPO_ROOT"; O="$R/gpurun_out"
W=${1:-3840}; H=${2:-2160}
i=0
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES GRBM_GUI_ACTIVE" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $grp -d $O/te_$i -o p -- python3 $R/tools/experiments/frames_only.py $W $H 20 phase 1 > /dev/null 2> $O/te_$i.log
done
cd "$R"
for j in $(seq 1 $i); do python3 tools/pmc_dump.py gpurun_out/te_$j k_tile_expand; rm -rf gpurun_out/te_$j; done
