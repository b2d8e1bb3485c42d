#!/bin/bash
# Regenerates reference dumps and fixtures (CONTAINER-ONLY, see build.sh / build_ocv.sh).
#   run_all.sh [CASE ...]            regenerate and (re)pack tests/golden/<case>.npz + manifest.json
#   run_all.sh --verify [CASE ...]   regenerate into oracle/_dumps only and compare every array's sha256 with the COMMITTED
#                                    manifest (verify.py): the check that the fixtures are what the reference produces
set -euo pipefail
HERE=$(cd "$(dirname "$0")" && pwd)
VERIFY=0
if [ "${1:-}" = "--verify" ]; then VERIFY=1; shift; fi
"$HERE/build_ocv.sh"
"$HERE/build.sh"
python3 "$HERE/make_inputs.py" "$@" | while read -r mode case; do
  "$HERE/../_ref/gen_golden" "$mode" "$HERE/../_dumps/cases/$case"
done
if [ $VERIFY = 1 ]; then python3 "$HERE/verify.py" "$@"; else python3 "$HERE/pack.py" "$@"; fi
