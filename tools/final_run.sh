#!/bin/bash
# The round's closing measurements, one gpurun call: the official line, ViT-H, the BASELINE configs through the drop-in
# table, then the profile collection (tools/collect_profiles.sh).  Results under gpurun_out/final/ and
# gpurun_out/profile_summary_vit_b/; tools/publish_profiles.sh RNN copies them into profiles/ under their tracked names.
set -o pipefail
mkdir -p gpurun_out/final
F=gpurun_out/final
timeout -k 10 400 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $F/bench_vit_b_b1.json 2> $F/bench_vit_b_b1.err && echo "official line ok" &&
timeout -k 10 300 python3 bench.py --steps 200 --warmup 10 --repeats 5 --no-cpu-baseline --no-abi-path --no-config-legs > $F/bench_vit_b_b1_steps200.json 2> $F/b200.err && echo "steps 200 ok" &&
timeout -k 10 400 python3 bench.py --model vit_h --steps 20 --warmup 5 --no-cpu-baseline --no-config-legs > $F/bench_vit_h_b1.json 2> $F/bench_vit_h.err && echo "vit_h ok" &&
timeout -k 10 300 python3 tools/bench_configs.py vit_b > $F/configs_vit_b.txt 2>&1 && echo "configs vit_b ok" &&
timeout -k 10 400 python3 tools/bench_configs.py vit_h > $F/configs_vit_h.txt 2>&1 && echo "configs vit_h ok" &&
timeout -k 10 300 python3 bench.py --gpus 2 --rehearse-gloo --steps 10 --warmup 2 --repeats 5 > $F/bench_gpus2_rehearsal.json 2> $F/rehearsal.err && echo "rehearsal ok" &&
timeout -k 10 300 python3 tools/power_kernels.py vit_b 3 > $F/power_kernels_vit_b.txt 2>&1 && echo "power ok" &&
timeout -k 10 200 python3 tools/defer_probe.py > $F/defer_probe.txt 2>&1 && echo "defer probe ok" &&
bash tools/collect_profiles.sh vit_b
