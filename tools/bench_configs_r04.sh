#!/bin/bash
# The exact bench.py lines for BASELINE.json's other configurations (run on the GPU box): each full JSON line is kept under
# gpurun_out/r04_bench_<tag>.json and copied into profiles/ (evidence for C2 / C4 / C5, not scratch citations).
cd $GRAFT_REPO_ROOT
run() { tag=$1; shift; timeout -k 5 900 python bench.py --cpu-steps 0 --no-extras "$@" 2> gpurun_out/r04_bench_$tag.err | tail -1 > gpurun_out/r04_bench_$tag.json; python -c "
import json,sys
d=json.load(open('gpurun_out/r04_bench_$tag.json')); print('$tag', '$*', '| ms/step', d['ms_per_step'], '| value', d['value'], '| roof', d['roofline']['kernel'], d['roofline']['frac'])"; }
run c2_b4_ddpm250_b64       --mode sample --graph --model DiffMa-B/4 --batch-per-gpu 64 --steps 50 --warmup 5
run c2_b4_ddpm250_b8        --mode sample --graph --model DiffMa-B/4 --batch-per-gpu 8 --steps 50 --warmup 5
run c3_l2_train_b8_graph    --model DiffMa-L/2 --batch-per-gpu 8 --steps 20 --graph
run c3_l2_train_b8_eager    --model DiffMa-L/2 --batch-per-gpu 8 --steps 20
run c4_xl2_mamba2_b64_graph --model DiffMa-XL/2 --use-mamba2 --batch-per-gpu 64 --steps 8 --warmup 2 --graph
run c4_xl2_mamba2_b64_eager --model DiffMa-XL/2 --use-mamba2 --batch-per-gpu 64 --steps 8 --warmup 2
run c5_xxl2_ddim50_b64      --mode sample --graph --sampler ddim50 --model DiffMa-XXL/2 --batch-per-gpu 64 --steps 30 --warmup 5
run c5_xxl2_ddim50_b8       --mode sample --graph --sampler ddim50 --model DiffMa-XXL/2 --batch-per-gpu 8 --steps 30 --warmup 5
run c3_l2_train_b1_graph    --model DiffMa-L/2 --batch-per-gpu 1 --steps 40 --warmup 5 --graph
run route_a_b512            --model DiffMa-L/2 --route-a
run route_a_b64             --model DiffMa-L/2 --route-a --batch-per-gpu 64
run native_b64              --model DiffMa-L/2 --batch-per-gpu 64
