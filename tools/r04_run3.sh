#!/usr/bin/env bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out/r04c
export TMPDIR=/tmp
p=29611
for w in a2a_eager all_eager all_captured a2a_captured; do
  p=$((p+1))
  timeout 300 python tools/r04_rccl_probe.py $w $p > gpurun_out/r04c/probe_$w.log 2>&1
  echo "$w rc=$?"; grep -E "OK|passes|Error|error|Segmentation|File " gpurun_out/r04c/probe_$w.log | tail -8
done
