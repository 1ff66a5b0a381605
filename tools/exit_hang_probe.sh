#!/bin/bash
# tools/exit_hang_probe.sh -- does the drop-in leave after SIGINT? Runs it, interrupts it, and if it is still there
# 6 s later shows where its threads sit, then kills it.
cd "$(dirname "$0")/.."
flags=${1:-"-m l -s 16000000 --filter"}
HVK_BATCH=32 HVK_SHIM_STATS=1 oracle/_ref/hacktv_hvk $flags -o /dev/null test > /tmp/probe.log 2>&1 &
pid=$!
sleep ${2:-5}
kill -INT $pid
for i in $(seq 12); do sleep 0.5; kill -0 $pid 2>/dev/null || break; done
if kill -0 $pid 2>/dev/null; then
  echo "STILL RUNNING after SIGINT + 6 s"
  tail -4 /tmp/probe.log
  for t in /proc/$pid/task/*; do echo "$(basename $t) $(cat $t/comm) wchan=$(cat $t/wchan 2>/dev/null) state=$(grep State $t/status)"; done
  kill -USR1 $pid; sleep 1; echo "--- worker backtrace"; grep -A40 "Caught signal" /tmp/probe.log | tail -40
  which gdb >/dev/null 2>&1 && timeout 20 gdb -p $pid -batch -ex "thread apply all bt 12" 2>/dev/null | grep -v "^\[New\|^Using\|^warning" | head -80
  kill -KILL $pid
else
  echo "left by itself"; tail -3 /tmp/probe.log
fi
