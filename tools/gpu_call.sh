#!/bin/bash
# Helper for gpurun calls: run steps one after another, each under its own time limit; a step that is KILLED at its limit (or
# dies on a signal) ends the call -- no further GPU step is started after a hang.  A step that merely fails (tests red) does not.
#   step <seconds> <log name> <command ...>
step() {
  local lim=$1 log=$2; shift 2
  echo "== $log: $*" | tee -a gpurun_out/progress.log
  local t0=$(date +%s)
  timeout -k 10 "$lim" "$@" > "gpurun_out/$log" 2> "gpurun_out/$log.err"
  local rc=$?
  echo "== $log rc=$rc $(( $(date +%s) - t0 )) s" | tee -a gpurun_out/progress.log
  if [ $rc -ge 124 ]; then echo "step killed: stopping the call" | tee -a gpurun_out/progress.log; exit $rc; fi
  return 0
}
