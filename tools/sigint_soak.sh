#!/bin/bash
# tools/sigint_soak.sh [n] -- the drop-in interrupted n times per mode (SIGINT after 3 s, SIGKILL 4 s later if it has not
# gone): every run has to end by itself (timeout reports 124, not 137; the statistics line printed)
cd "$(dirname "$0")/.."
N=${1:-6}
bad=0
for flags in "-m l -s 16000000 --filter" "-m i -s 16000000 --filter" "-m i -s 16000000 --filter --noaudio"; do
  for i in $(seq $N); do
    out=$(HVK_BATCH=32 HVK_SHIM_STATS=1 timeout -k 4 -s INT 3 oracle/_ref/hacktv_hvk $flags -o /dev/null test 2>&1)
    rc=$?
    if [ $rc -ne 124 ] || ! echo "$out" | grep -q "frames in"; then bad=$((bad + 1)); echo "run $i of '$flags': exit $rc"; echo "$out" | tail -3; fi
  done
done
echo "sigint soak: $bad bad run(s)"
[ $bad -eq 0 ]
