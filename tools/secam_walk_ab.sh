#!/bin/bash
# tools/secam_walk_ab.sh -- the three kernels that can walk a SECAM block's lines (HVK_SECAM_WALK=0 hvk_k_secam_chain,
# 1 hvk_k_secam_walk<0>: table, 2 hvk_k_secam_walk<1>: FM steps computed), on the test card, on noisy pictures that change
# and on a new picture per frame, 512-frame blocks; then the per-kernel times of the engine's own choice (rocprofv3).
set -u
OUT=$PWD/gpurun_out/secam_ab
mkdir -p "$OUT"
export TMPDIR=/tmp
for kind in card noisy new; do
	for w in 0 1 2 auto; do
		if [ $w = auto ]; then unset HVK_SECAM_WALK; else export HVK_SECAM_WALK=$w; fi
		echo "HVK_SECAM_WALK=$w: $(timeout 200 python tools/secam_blocks.py 512 $kind 8 2>&1 | tail -1)"
	done
done | tee "$OUT/ab.txt"
unset HVK_SECAM_WALK
cd /tmp
for kind in card noisy new; do
	timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/$kind" -o p -- python $OLDPWD/tools/secam_blocks.py 512 $kind 8 > "$OUT/$kind.log" 2>&1
	f=$(find "$OUT/$kind" -name '*kernel_stats.csv' | head -1)
	[ -n "$f" ] && cp "$f" "$OUT/secam_${kind}_kernel_stats.csv" && head -8 "$f" | cut -c1-150
done
