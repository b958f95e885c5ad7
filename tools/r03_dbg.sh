#!/bin/bash
# long concurrent decoder stress + decoder timings (tools/decoder_stress.py, tools/r03_dec.sh)
timeout -k 10 500 python3 tools/decoder_stress.py 4 80000 state 2>&1 | tail -4 | cut -c1-300
timeout -k 10 200 python3 tools/decoder_stress.py 4 120 masks 2>&1 | tail -3 | cut -c1-300
