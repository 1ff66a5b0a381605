#!/bin/bash
# tools/dropin_devnull.sh [seconds] -- the drop-in binary (the reference's main() and file sink over the shim) writing to /dev/null,
# --noaudio and with sound: the shim's own account at exit (HVK_SHIM_STATS). Run on the GPU box.
S=${1:-6}
for b in 32 128; do
for f in "--noaudio" ""; do
	echo "== HVK_BATCH=$b hacktv_hvk -m i -s 16000000 --filter $f -o /dev/null test ($S s)"
	HVK_SHIM_STATS=1 HVK_BATCH=$b timeout -s INT $S oracle/_ref/hacktv_hvk -m i -s 16000000 --filter $f -o /dev/null test 2>&1 | grep "hacktv-amd" | cut -c1-400
done
done
