# does the warp kernel care whether the sources' rows are 256-byte aligned?  (3840 x 3 = 45 x 256 bytes: 45 us; 3836 / 3844: 57 us)
cd /tmp; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT
for sz in "1792 1080" "1856 1080" "1920 1080" "1984 1080" "2048 1080" "3584 2160" "3712 2160" "3840 2160" "3968 2160" "4096 2160"; do w=${sz% *}
timeout 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/wa_$w -o t -- python3 $R/tools/experiments/frames_only.py $sz 60 chain 2 > /dev/null 2>&1
echo "$sz (row $((w*3)) B = $(python3 -c "print($w*3/256)") x 256): $(python3 $R/tools/rocprof_summary.py $R/gpurun_out/wa_$w/t_results.db | grep -E "k_warp_bin|k_collapse_level<true>|k_unsharp_|k_pyrdown_level<true>" | awk -F'|' -v px=$((w*${sz#* })) '{printf "%s %s (%.2f ns/kpx); ", $2, $5, $5*1e6/px}')"
rm -rf $R/gpurun_out/wa_$w
done
