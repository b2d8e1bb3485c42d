#!/bin/bash
# Regenerates reference dumps and fixtures (CONTAINER-ONLY, see build.sh).  usage: run_all.sh [CASE ...]
set -euo pipefail
HERE=$(cd "$(dirname "$0")" && pwd)
"$HERE/build.sh"
python3 "$HERE/make_inputs.py" "$@" | while read -r mode case; do
  "$HERE/../_ref/gen_golden" "$mode" "$HERE/../_dumps/cases/$case"
done
python3 "$HERE/pack.py" "$@"
