#!/bin/bash
# the wrong-element hunt: several copies of the tuning build (lib/libdlimgedit_var*.so), same box, same stress
for v in "$@"; do echo "== $v"; DLIMGEDIT_TUNING_LIB=libdlimgedit_$v.so timeout -k 10 400 python3 tools/decoder_stress.py 4 60000 state 2>&1 | tail -7 | cut -c1-260; done
