# frame copies on event-free streams of their own (POPPY_HIP_DL_STREAMS=1) against the download stream + event records: the pool of six with the writer,
# in a process on the image's runtime (copies on the SDMA engines) and in one that has torch's runtime (copies as blit kernels)
cd $GRAFT_REPO_ROOT
for rep in 1 2; do
echo "events,  sdma:  $(timeout 300 python3 tools/experiments/pool_nowriter.py 2 6 36 1 6 | tail -1)"
echo "streams, sdma:  $(POPPY_HIP_DL_EVENTS= timeout 300 python3 tools/experiments/pool_nowriter.py 2 6 36 1 6 | tail -1)"
echo "events,  torch: $(POOL_WITH_TORCH=1 timeout 300 python3 tools/experiments/pool_nowriter.py 2 6 36 1 6 | tail -1)"
echo "streams, torch: $(POOL_WITH_TORCH=1 POPPY_HIP_DL_EVENTS= timeout 300 python3 tools/experiments/pool_nowriter.py 2 6 36 1 6 | tail -1)"
echo "events,  one context: $(timeout 300 python3 tools/experiments/pool_nowriter.py 4 1 6 1 6 | tail -1)"
echo "streams, one context: $(POPPY_HIP_DL_EVENTS= timeout 300 python3 tools/experiments/pool_nowriter.py 4 1 6 1 6 | tail -1)"
done
POPPY_HIP_DL_EVENTS= timeout 900 python3 -m pytest tests/test_gpu_sequences.py -x -q -m gpu -k "cfg1 or cfg2 or writer or download or cfg5" 2>&1 | tail -2
