#!/bin/bash
# kernel durations of the attention probe under rocprofv3 --kernel-trace for each ablation of the ping-pong kernel
cd /tmp && export TMPDIR=/tmp
R="${GRAFT_REPO_ROOT:?GRAFT_REPO_ROOT is not set (run through gpurun)}"
for a in ${ABLS:-0 1 2 3}; do
  rm -rf /tmp/kt_$a
  DLIMGEDIT_ATTN_PP_ABLATE=$a timeout -k 10 90 rocprofv3 --kernel-trace -d /tmp/kt_$a -o k -- python3 $R/tools/attn_probe.py global 12 64 5 > /tmp/kt_$a.log 2>&1
  echo "ablate=$a rc=$?"
  f=$(ls /tmp/kt_$a/*/k_results.db /tmp/kt_$a/k_results.db 2>/dev/null | head -1)
  python3 $R/tools/kernel_stats.py $f 3 2>&1 | grep -i "attention" | cut -c1-90
done
