# how often the pooled headline lands in its low state: N runs of the headline-only bench [extra bench arguments]
cd $GRAFT_REPO_ROOT
n=${1:-12}; shift
for i in $(seq 1 $n); do
  python3 bench.py --headline-only --steps 25 "$@" 2>/dev/null | python3 -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print(d['value'], d.get('value_unselected'), d['pool_selection']['candidates_ms_per_batch'], d['pool_selection']['kept'], d['step_ms']['median'])"
done
