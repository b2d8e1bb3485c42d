#!/bin/bash
# In-process A/B of the wave-priority stagger, kernel by kernel: the experiments build (python -m poppy_amd.build --experiments, CONTAINER) alternates the
# kernels of POPPY_STAGGER_AB launch by launch; one traced chained loop per size holds both forms.  usage: gpurun -- bash tools/experiments/stagger_ab.sh [mask]
mask=${1:-415}
cd /tmp && export TMPDIR=/tmp
R="$GRAFT_REPO_ROOT"; O="$R/gpurun_out/r06_stagger_ab"; mkdir -p $O
export POPPY_HIP_LIB=$R/poppy_amd/libpoppy_hip_experiments.so POPPY_STAGGER_AB=$mask
for sz in "1920 1080" "3840 2160" "2560 1440"; do set -- $sz
  timeout 300 rocprofv3 --kernel-trace -d $O/t_$1 -o t -- python3 $R/tools/experiments/frames_only.py $1 $2 60 chain 4 > $O/log_$1.txt 2>&1
  echo "== $1 x $2 (mask $mask)"; python3 $R/tools/experiments/stagger_ab.py $O/t_$1/t_results.db
done 2>&1 | tee $O/ab.txt
