#!/bin/bash
# tools/pmc_quick.sh TAG [env assignments...] -- one rocprofv3 --pmc pass of the instruction / wave counters
# over a short bench run, per-kernel averages printed (run on the GPU box)
set -u
TAG=${1:-q}; shift
OUT=$PWD/gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
BENCH="env $* python $PWD/bench.py"
cd /tmp
timeout 120 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_LDS --output-format csv -d "$OUT/pmc1" -o p -- $BENCH --steps 3 --warmup 1 --no-cpu-baseline --no-moving > "$OUT/pmc1.log" 2>&1
timeout 120 rocprofv3 --kernel-trace --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES --output-format csv -d "$OUT/pmc2" -o p -- $BENCH --steps 3 --warmup 1 --no-cpu-baseline --no-moving > "$OUT/pmc2.log" 2>&1
cd "$OLDPWD"
python tools/pmc_summary.py "$OUT" "$TAG" | head -c 1500
