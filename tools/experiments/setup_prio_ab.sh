#!/bin/bash
# Round 6: (a) the finer host timing of the set-up's tail (POPPY_SETUP_TIMING), (b) the second image's chain on a high-priority stream (POPPY_SETUP_PRIO) A/B,
# device-resident pairs, 25 set-ups each, three alternations.   gpurun -- bash tools/experiments/setup_prio_ab.sh
# (POPPY_SETUP_PRIO existed only in the build this was run on — hipStreamCreateWithPriority for the second chain's stream in pair_setup.cpp —: no difference, removed; profiles/r06_setup_tail.txt)
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/setup_prio.txt; : > $O
POPPY_SETUP_TIMING=1 python3 tools/experiments/setup_device_time.py synthetic 6 2>&1 | tail -32 >> $O
for r in 1 2 3; do
  echo "default:  $(python3 tools/experiments/setup_device_time.py synthetic 25 2>&1 | tail -1)" >> $O
  echo "aux high: $(POPPY_SETUP_PRIO=1 python3 tools/experiments/setup_device_time.py synthetic 25 2>&1 | tail -1)" >> $O
done
echo "photo default:  $(python3 tools/experiments/setup_device_time.py photo 25 2>&1 | tail -1)" >> $O
echo "photo aux high: $(POPPY_SETUP_PRIO=1 python3 tools/experiments/setup_device_time.py photo 25 2>&1 | tail -1)" >> $O
cat $O
