# the headline on other pool shapes (contexts x pairs per step), --headline-only, one box
cd $GRAFT_REPO_ROOT
for rep in 1 2; do for cp in "3 6" "4 8" "2 4" "3 9" "4 4" "6 6"; do set -- $cp
  echo "contexts $1 pairs $2: $(python3 bench.py --headline-only --contexts $1 --pairs $2 --steps 30 2>/dev/null | python3 -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print(d['value'], d.get('value_unselected'), d['ms_per_step'])")"
done; done
