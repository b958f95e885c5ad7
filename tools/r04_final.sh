#!/bin/bash
# round 4 closing run: the official line, steady state, batch 8, ViT-H, the BASELINE configs through the ABI, the hazard
# probe, and the profile collection of the bench command (one gpurun call)
set -o pipefail
O=gpurun_out/r04_final; mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc $?"; tail -3 $O/pytest.log
timeout -k 10 300 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_vit_b_b1.json 2> $O/bench.err; echo "official rc $?"
python tools/show_bench.py r04_final/bench_vit_b_b1 | cut -c1-300
timeout -k 10 300 python bench.py --steps 200 --warmup 10 --repeats 5 --no-abi-path --no-cpu-baseline > $O/bench_vit_b_b1_steps200.json 2>> $O/bench.err; echo "steps200 rc $?"
python tools/show_bench.py r04_final/bench_vit_b_b1_steps200 | cut -c1-120
timeout -k 10 300 python bench.py --steps 10 --batch 8 --no-abi-path --no-cpu-baseline > $O/bench_vit_b_b8.json 2>> $O/bench.err; echo "b8 rc $?"
python tools/show_bench.py r04_final/bench_vit_b_b8 | cut -c1-120
timeout -k 10 400 python bench.py --model vit_h --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_vit_h_b1.json 2>> $O/bench.err; echo "vit_h rc $?"
python tools/show_bench.py r04_final/bench_vit_h_b1 | cut -c1-300
timeout -k 10 300 python tools/bench_configs.py vit_b > $O/configs_vit_b.txt 2>&1; echo "configs vit_b rc $?"; tail -8 $O/configs_vit_b.txt
timeout -k 10 400 python tools/bench_configs.py vit_h > $O/configs_vit_h.txt 2>&1; echo "configs vit_h rc $?"; tail -8 $O/configs_vit_h.txt
timeout -k 10 200 tools/_bin/pkfma_hazard 3 > $O/pkfma_hazard.txt 2>&1; echo "hazard rc $?"
bash tools/collect_profiles_r04.sh vit_b > $O/collect_b.log 2>&1; echo "collect rc $?"; tail -3 $O/collect_b.log
