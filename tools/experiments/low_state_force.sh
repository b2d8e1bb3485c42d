# the low state against the per-image choice of median kernel: forced to the column histograms (1), to the lane-per-column kernel (0), or by the device count (unset)
cd $GRAFT_REPO_ROOT
run() { python3 bench.py --headline-only --steps 20 2>/dev/null | python3 -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('   ', d['value'], d.get('value_unselected'), d['pool_selection']['candidates_ms_per_batch'], d['step_ms']['median'])"; }
for f in 1 0; do echo "POPPY_MED_COLS_FORCE=$f"; export POPPY_MED_COLS_FORCE=$f; for i in $(seq 1 ${1:-8}); do run; done; done
