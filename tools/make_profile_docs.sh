#!/bin/bash
# Turns the raw outputs of tools/profile_round.sh <tag> (gpurun_out/<tag>_*) into the files kept under profiles/:
#   profiles/<tag>_trace.md, profiles/<tag>_pmc.md, profiles/<tag>_bench.json, profiles/<round>_warp_pmc.json, profiles/<round>_warp_facts.json
# Usage (in the build container, after `gpurun -- bash tools/profile_round.sh <tag>`):  bash tools/make_profile_docs.sh <tag>
tag=${1:?tag}
cd "$(dirname "$0")/.."
RID="${POPPY_RUN_ID:-unstamped (formatted outside tools/profile_round.sh)}"
F='k_median\|k_gauss23\|k_acc_\|k_pad_cols\|k_orb_\|k_fast\|k_mog2\|k_dft\|k_spectrum\|k_equalize\|k_apply_lut\|k_fg_mask\|k_sep_\|k_unsharp1\|k_u8_to\|k_bgr2\|k_harris\|k_ic_angle\|k_pad_complex\|rocclr\|k_gabor\|k_orb'
T=profiles/${tag}_trace.md
{ echo "# Final build of the round: kernel traces (rocprofv3 --kernel-trace --stats), MI355X"; echo
  echo "Run id: \`$RID\` — hostname, UTC time and tag of the ONE \`gpurun\` call every file of this tag was written by (\`${tag}_trace.md\`, \`${tag}_pmc.md\`, \`${tag}_bench.json\`, \`${tag%%_*}_warp_pmc.json\`, \`${tag%%_*}_warp_facts.json\`: the same id in each)."; echo
  echo "Collected by \`tools/profile_round.sh ${tag}\` (\`gpurun -- bash tools/profile_round.sh ${tag}\`), formatted by \`tools/make_profile_docs.sh\`.  Four runs: the default bench command's timed workload (the default pool: pair set-up + 60 chained frames + writer per pair, torch in the process: the frame downloads show up as \`__amd_rocclr_copyBuffer\` blit kernels, see r02_notes.md 6.8), the chained frame loop alone at 1080p and at 4K (\`tools/experiments/frames_only.py W H 60 chain 3\`, one context, frames left in HBM; the first call includes one pair set-up), and the pair set-up (\`tools/experiments/pair_begin_time.py\`: 4 set-ups + one 60-frame sequence)."; echo
  python3 tools/rocprof_summary.py gpurun_out/${tag}_bench_trace/t_results.db --title "python3 bench.py --steps 4 --warmup 1 --headline-only (1080p, default pool)" | head -48; echo
  python3 tools/rocprof_summary.py gpurun_out/${tag}_chain_1920/t_results.db --title "frames_only.py 1920 1080 60 chain 3" --by-grid | grep -v "$F"; echo
  python3 tools/rocprof_summary.py gpurun_out/${tag}_chain_3840/t_results.db --title "frames_only.py 3840 2160 60 chain 3" --by-grid | grep -v "$F"; echo
  python3 tools/rocprof_summary.py gpurun_out/${tag}_setup_trace/t_results.db --title "pair_begin_time.py (4 pair set-ups at 1080p + 60 chained frames with the writer)" | head -62; } > $T
python3 tools/pmc_json.py ${tag} > /tmp/pmc_json.out
P=profiles/${tag}_pmc.md
{ echo "# Final build of the round: HBM traffic per launch (rocprofv3 --pmc FETCH_SIZE, --pmc WRITE_SIZE; separate passes, --pmc only)"; echo
  echo "Run id: \`$RID\`."; echo
  echo "Command per pass: \`rocprofv3 --pmc <counter> -- python3 tools/experiments/frames_only.py W H 60 chain 1\` (\`tools/profile_round.sh ${tag}\`).  FETCH_SIZE is doubled for wide coalesced reads as MI355X_MICROARCH.md prescribes; the in-run calibration row is \`k_gray_inv\` (reads exactly 12 B/px, writes 4: raw 6.00 / x2 12.00 / 4.00).  \`profiles/${tag%%_*}_warp_pmc.json\` (written by \`tools/pmc_json.py ${tag}\`) holds the per-launch bytes \`bench.py\` quotes as \`roofline.traffic\`, with a hash of the kernel sources they were taken from."; echo
  for sz in "1920 1080" "3840 2160"; do set -- $sz
    echo "## $1x$2"; echo
    python3 tools/pmc_traffic.py gpurun_out/${tag}_fetch_$1/f_results.db gpurun_out/${tag}_write_$1/w_results.db --px $1*$2 | grep -v "$F"; echo
    grep "$1x$2 k_warp_bin" /tmp/pmc_json.out | sed 's/^/`/; s/ grid / (grid /; s/: read /): read (x2) /; s/$/; the images alone are 12 B\/px (c1 3 + c2 3 in, tr1 3 + tr2 3 out), the rest are the id bytes (1 B\/px) and the record slots of the tiles.  No id map, no blend mask./'; echo
  done; } > $P
[ -f gpurun_out/${tag}_bench.json ] && cp gpurun_out/${tag}_bench.json profiles/${tag}_bench.json      # (tools/profile_round.sh runs the bench AFTER this script and copies it itself)
python3 tools/warp_facts.py ${tag}
wc -l $T $P
