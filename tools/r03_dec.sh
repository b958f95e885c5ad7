#!/bin/bash
# decoder iteration loop: parity of the decoder tests, latency of compute_mask, kernel timeline of one call
set -e
R=$PWD; out=$R/gpurun_out/dec; mkdir -p $out
python -m pytest tests/test_gpu_e2e.py -x -q -k "logits or multi_mask or prompt or batch" > $out/tests.txt 2>&1 || { tail -30 $out/tests.txt; exit 1; }
tail -2 $out/tests.txt
python3 tools/decode_probe.py vit_test 200 1 > $out/lat.txt
python3 tools/decode_probe.py vit_test 100 2 >> $out/lat.txt
python3 tools/decode_probe.py vit_test 100 5 >> $out/lat.txt
python3 tools/decode_probe.py vit_test 50 16 >> $out/lat.txt
cat $out/lat.txt
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/tr && rocprofv3 --kernel-trace -d /tmp/tr -o tr -- python3 $R/tools/decode_probe.py vit_test 20 1 > /dev/null 2>&1
f=$(find /tmp/tr -name "*results.db" | head -1)
python3 $R/tools/decode_timeline.py $f > $out/timeline.txt 2>&1
cat $out/timeline.txt
