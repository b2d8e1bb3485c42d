# the pooled headline's low state against CPU placement: N runs of the headline-only bench pinned to each socket's CPUs
cd $GRAFT_REPO_ROOT
lscpu | grep -E "NUMA|Socket|Core|Thread|Model name" | head -12
cat /sys/class/drm/card*/device/numa_node 2>/dev/null | tr '\n' ' '; echo " <- numa_node of the GPU(s)"
run() { python3 bench.py --headline-only --steps 20 2>/dev/null | python3 -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('   ', d['value'], d.get('value_unselected'), d['pool_selection']['candidates_ms_per_batch'], d['step_ms']['median'])"; }
N0=$(cat /sys/devices/system/node/node0/cpulist); N1=$(cat /sys/devices/system/node/node1/cpulist 2>/dev/null)
echo "node0 cpus $N0"; for i in $(seq 1 ${1:-8}); do taskset -c $N0 bash -c "$(declare -f run); run"; done
if [ -n "$N1" ]; then echo "node1 cpus $N1"; for i in $(seq 1 ${1:-8}); do taskset -c $N1 bash -c "$(declare -f run); run"; done; fi
