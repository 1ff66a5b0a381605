#!/bin/bash
# tools/host_asan.sh [cases] -- the library's host half in C (hvk_tables.c, hvk_audio.c, hvk_secam.c, hvk_tail.c, hvk_presets.c: tables,
# the serial sound / SECAM / tail chains) compiled with gcc's AddressSanitizer + UndefinedBehaviorSanitizer into /tmp/libhvk_asan.so
# (the HIP objects as ever; sanitizers on the GPU are not available on this pool) and run without a device: the host-path and ABI tests,
# tools/refusal_rate.py's random opens, tools/fuzz_oracle_ref.py's engine leg (tables and the sound pre-pass against the oracle's).
# No GPU needed. Any report ends the run (halt_on_error).
set -e
cd "$(dirname "$0")/.."
N=${1:-300}
make -s -C hacktv_amd/csrc -j8 CC="gcc -fsanitize=address,undefined -fno-omit-frame-pointer -fno-sanitize-recover=undefined" B=/tmp/build_asan OUT=/tmp/libhvk_asan.so
export ASAN_OPTIONS=detect_leaks=0:halt_on_error=1 HVK_LIB=/tmp/libhvk_asan.so
export LD_PRELOAD="$(gcc -print-file-name=libasan.so) $(gcc -print-file-name=libubsan.so)"
python -m pytest tests/test_host_path.py tests/test_abi.py -x -q 2>&1 | tail -1
python tools/refusal_rate.py $((N * 5)) 9 2>&1 | grep "configurations drawn"
FUZZ_FSC_PIXELS=1 python tools/fuzz_oracle_ref.py $N 7100 900 6 2>&1 | tail -1
