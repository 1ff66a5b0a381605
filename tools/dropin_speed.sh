#!/bin/bash
# tools/dropin_speed.sh -- end-to-end speed of the reference CLI with its own engine (hacktv_ref)
# and with the MI355X engine behind the video.h shim (hacktv_hvk): same main(), same test source,
# same file sink writing int16 IQ to /dev/null. Seconds of signal per second of wall clock.
cd "$(dirname "$0")/.."
S=${1:-20}            # seconds of 16 Msps signal
N=$((S * 16000000 * 4))
# steady state of the shim, start-up (device, tables: ~0.5 s) left out, sink = /dev/null without a pipe in between:
# HVK_SHIM_STATS prints frames / time from the first line on and where both threads' time went
if [ -n "${STEADY:-}" ]; then
  for flags in "-m i -s 16000000 --filter" "-m i -s 16000000 --filter --noaudio" "-m l -s 16000000 --filter"; do
    for extra in "" "HVK_SHIM_PAGEABLE=1"; do
      echo "== hacktv_hvk $flags $extra"
      env HVK_BATCH=32 HVK_SHIM_STATS=1 $extra timeout -k 5 -s INT ${STEADY} oracle/_ref/hacktv_hvk $flags -o /dev/null test 2>&1 | grep "hacktv-amd"
    done
  done
  exit 0
fi
for b in oracle/_ref/hacktv_ref "env HVK_BATCH=32 oracle/_ref/hacktv_hvk"; do
  for flags in "-m i -s 16000000 --filter" "-m i -s 16000000 --filter --noaudio" "-m l -s 16000000 --filter"; do
    t0=$(date +%s.%N)
    $b $flags -o - test 2>/dev/null | head -c $N > /dev/null
    t1=$(date +%s.%N)
    python3 -c "print('%-44s %-40s %6.1f Msamples/s  (%.1f x real time)' % ('$b'.split('/')[-1], '$flags', $S*16/($t1-$t0), $S/($t1-$t0)))"
  done
done
