#!/bin/bash
# tools/ref_asan.sh [ref_random_check.py arguments ...] -- TEST INFRASTRUCTURE. The unmodified reference's sources, compiled where they
# lie (/root/reference/src) with -fsanitize=address into /tmp/hvk_ref_asan/libhacktv_ref_asan.so (nothing of it enters the repository),
# and tests/ref_random_check.py run on it (tests/refprobe.py: HVK_REF_LIB): which of the reference's accesses leave their buffers on a
# given configuration -- where its output depends on what the allocator put behind them, a comparison cannot be had.
#   tools/ref_asan.sh ntsc_sv_f_16_27
#   tools/ref_asan.sh '@{"name": "fz60221_2938", "setup": [...]}'
set -e
REF=${REF:-/root/reference/src}
R=$(cd "$(dirname "$0")/.." && pwd)
OUT=/tmp/hvk_ref_asan
mkdir -p $OUT
if [ ! -f $OUT/libhacktv_ref_asan.so ]; then
	F="-O1 -g -fsanitize=address -fsanitize-recover=address -fno-omit-frame-pointer -pthread -fPIC -I$REF"
	for f in common fir vbidata teletext wss video fifo mac dance eurocrypt videocrypt videocrypts syster syster-ca acp vits vitc nicam728 sis av av_test rf rf_file spdif cc608; do
		gcc $F -DVERSION=\"asan\" -c $REF/$f.c -o $OUT/$f.o 2>/dev/null &
	done
	wait
	gcc $F -c $R/oracle/ref_stubs.c -o $OUT/ref_stubs.o
	gcc $F -I$R/hacktv_amd/csrc/shim -c $R/oracle/ref_probe.c -o $OUT/ref_probe.o
	gcc -shared -fsanitize=address -o $OUT/libhacktv_ref_asan.so $OUT/*.o -lm -pthread
fi
HVK_REF_LIB=$OUT/libhacktv_ref_asan.so LD_PRELOAD=$(gcc -print-file-name=libasan.so) ASAN_OPTIONS=detect_leaks=0:halt_on_error=0 \
	python $R/tests/ref_random_check.py "$@" > $OUT/out.txt 2>&1 || true
grep -A3 "ERROR: AddressSanitizer" $OUT/out.txt | grep -E "READ|WRITE|#0" | awk '{$1=$1};1' | cut -c1-160 | sort | uniq -c | sort -rn
tail -1 $OUT/out.txt
