# the 480-frame phase-mode job with a writer: slots on the context's three compute streams (round 3's answer to copies blocking a queue) against a stream per slot, now that copies block nothing;
# slots / ring sizes; torch's runtime and the image's
cd $GRAFT_REPO_ROOT
for t in 0 1; do
  if [ $t = 1 ]; then PRE="import torch;"; else PRE=""; fi
  for env in "X=1" "POPPY_PHASE_OWN_STREAMS=1" "POPPY_HIP_SLOTS=6 POPPY_HIP_RING=4" "POPPY_HIP_SLOTS=6 POPPY_HIP_RING=4 POPPY_PHASE_OWN_STREAMS=1" "POPPY_HIP_SLOTS=8 POPPY_HIP_RING=6 POPPY_PHASE_OWN_STREAMS=1" "POPPY_HIP_DL_EVENTS=1"; do
    echo "torch=$t $env: $(env $env python3 -c "$PRE exec(open('tools/experiments/job480.py').read())" 2>/dev/null | tail -1)"
  done
done
