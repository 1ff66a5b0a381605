#!/bin/bash
# tools/ref_msan.sh <reference CLI arguments ...> -- TEST INFRASTRUCTURE. The unmodified reference CLI compiled where its sources lie
# (/root/reference/src) with clang's MemorySanitizer into /tmp/hvk_ref_msan/hacktv_msan (nothing of it enters the repository) and run
# for a few seconds on the arguments given, output to /dev/null: every place where the reference USES a value it never wrote, with the
# allocation it came from. Where such a value reaches the output, the reference's output is what the allocator handed out -- zeros in
# the CLI's fresh heap, which is what the oracle and the engine assume.
#   tools/ref_msan.sh -m l -s 16000000 --secam-field-id --acp
set +e
REF=${REF:-/root/reference/src}
R=$(cd "$(dirname "$0")/.." && pwd)
CL=${CLANG:-/opt/rocm/lib/llvm/bin/clang}
OUT=/tmp/hvk_ref_msan
mkdir -p $OUT
if [ ! -x $OUT/hacktv_msan ]; then
	F="-O1 -g -fsanitize=memory -fsanitize-recover=memory -fsanitize-memory-track-origins=2 -fno-omit-frame-pointer -pthread -I$REF"
	for f in hacktv common fir vbidata teletext wss video fifo mac dance eurocrypt videocrypt videocrypts syster syster-ca acp vits vitc nicam728 sis av av_test rf rf_file spdif cc608; do
		$CL $F -DVERSION=\"msan\" -c $REF/$f.c -o $OUT/$f.o > /dev/null 2>&1 &
	done
	wait
	$CL $F -c $R/oracle/ref_stubs.c -o $OUT/ref_stubs.o
	$CL -fsanitize=memory -o $OUT/hacktv_msan $OUT/*.o -lm -pthread
fi
MSAN_OPTIONS=halt_on_error=0 timeout -k 1 ${SECONDS_TO_RUN:-8} $OUT/hacktv_msan "$@" -o /dev/null test > $OUT/run.log 2>&1
echo "$(grep -c 'WARNING: MemorySanitizer' $OUT/run.log) reports. Places of use:"
grep -A2 "WARNING: MemorySanitizer" $OUT/run.log | grep "#0" | sed 's/0x[0-9a-f]* //' | awk '{$1=$1};1' | sort | uniq -c | sort -rn | head -12
echo "Allocations the values came from:"
grep -A3 "created by" $OUT/run.log | grep -E "#[12] .* in " | grep -v "interceptors" | sed 's/0x[0-9a-f]* //' | awk '{$1=$1};1' | sort | uniq -c | sort -rn | head -8
