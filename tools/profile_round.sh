#!/bin/bash
# tools/profile_round.sh TAG -- the measurement set kept under profiles/ (run on the GPU box via gpurun):
#   gpurun --timeout 600 -- 'bash tools/profile_round.sh r01c'
# 1. python bench.py (un-profiled; HIP-event kernel times, cpu_baseline)      -> gpurun_out/TAG/bench.json
# 2. rocprofv3 --kernel-trace --stats of a short bench run                   -> gpurun_out/TAG/stats/
# 3. four separate --pmc passes (never combined with trace domains other than
#    --kernel-trace; TA_/TCP_ derived counters hang on this pool: not used)  -> gpurun_out/TAG/pmcN/
# 4. tools/pmc_summary.py folds 2-3 into TAG_kernel_stats.csv / TAG_pmc_counters.json / traffic.json
# 5. rocprofv3 --kernel-trace --stats of tools/secam_blocks.py (card, noisy)  -> TAG_secam_*_kernel_stats.csv
set -u
TAG=${1:-r01x}
BENCH_FLAGS=${BENCH_FLAGS:-}
OUT=$PWD/gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
BENCH="python $PWD/bench.py"

timeout 900 $BENCH $BENCH_FLAGS --detail-out "$OUT/bench_detail.json" > "$OUT/bench.json" 2> "$OUT/bench.err"
tail -c 600 "$OUT/bench.json"

cd /tmp
timeout 150 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -o p -- $BENCH --steps 200 --warmup 10 --no-cpu-baseline --no-configs > "$OUT/stats.log" 2>&1
i=0
for set in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_SALU" \
           "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR GRBM_GUI_ACTIVE"; do
	i=$((i + 1))
	timeout 90 rocprofv3 --kernel-trace --pmc $set --output-format csv -d "$OUT/pmc$i" -o p -- $BENCH --steps 20 --warmup 2 --settle 0 --no-cpu-baseline --no-configs > "$OUT/pmc$i.log" 2>&1
	echo "pmc pass $i ($set): exit $?"
done
# 5. the SECAM colour chain (hvk_secam.hip): blocks of 512 frames of the test card, of noisy pictures with the cells made per frame, and of
#    as many new pictures (planes made per frame as well)
for kind in card noisy new; do
	timeout 120 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/secam_$kind" -o p -- python $OLDPWD/tools/secam_blocks.py 512 $kind 8 > "$OUT/secam_$kind.log" 2>&1
	tail -1 "$OUT/secam_$kind.log"
	f=$(find "$OUT/secam_$kind" -name '*kernel_stats.csv' | head -1)
	[ -n "$f" ] && cp "$f" "$OUT/${TAG}_secam_${kind}_kernel_stats.csv"
done
# 5b. BASELINE config 4 without its sound, 128-frame blocks (bench.py's 4_secam_l_teletext_noaudio_device)
timeout 120 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/cfg4" -o p -- python $OLDPWD/tools/cfg4_blocks.py 128 20 > "$OUT/cfg4.log" 2>&1
tail -1 "$OUT/cfg4.log"
f=$(find "$OUT/cfg4" -name '*kernel_stats.csv' | head -1)
[ -n "$f" ] && cp "$f" "$OUT/${TAG}_cfg4_kernel_stats.csv"
# 6. pictures that change on every frame: the one kernel from the pixels (table levels) and the planes' two (computed levels)
timeout 120 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/moving" -o p -- env HVK_PATHS=0 HVK_CHUNKS=0 python $OLDPWD/tools/prep_speed.py 64 > "$OUT/moving.log" 2>&1
grep " i " "$OUT/moving.log" | cut -c1-220
f=$(find "$OUT/moving" -name '*kernel_stats.csv' | head -1)
[ -n "$f" ] && cp "$f" "$OUT/${TAG}_moving_kernel_stats.csv"
cd "$OLDPWD"
python tools/pmc_summary.py "$OUT" "$TAG"
# 7. FM video through the drop-in binary: the phasor pass in the caller's thread (HVK_FM_SYNC=1) and on the engine's
bash tools/fm_dropin_speed.sh > "$OUT/${TAG}_fm_dropin.txt" 2>&1; cat "$OUT/${TAG}_fm_dropin.txt" | grep -v worker
