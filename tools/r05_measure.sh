#!/bin/bash
# Round-5 side measurements (run on the GPU box): pair-path crossover, Mamba-2 saturating batch + PMC, fp32 stand-alone scan forward PMC.
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out
ms() { python -c "import sys,json; print(json.loads(sys.stdin.read().strip().splitlines()[-1])['ms_per_step'])"; }
if [ "$1" = "pair" ] || [ -z "$1" ]; then
  OUT=gpurun_out/pair_crossover.txt
  echo "DiffMa-L/2 training step, 1 x MI355X, bf16, ms/step: paired mixers (default) vs DIFFMA_PAIR_MIXERS=0, eager and whole-step hipGraph" > $OUT
  echo "batch   eager_pair  eager_nopair  graph_pair  graph_nopair" >> $OUT
  for B in 16 32 64 128; do
    a=$(python bench.py --batch-per-gpu $B --cpu-steps 0 --no-extras --steps 15 2>/dev/null | ms)
    b=$(DIFFMA_PAIR_MIXERS=0 python bench.py --batch-per-gpu $B --cpu-steps 0 --no-extras --steps 15 2>/dev/null | ms)
    c=$(python bench.py --batch-per-gpu $B --cpu-steps 0 --no-extras --steps 15 --graph 2>/dev/null | ms)
    d=$(DIFFMA_PAIR_MIXERS=0 python bench.py --batch-per-gpu $B --cpu-steps 0 --no-extras --steps 15 --graph 2>/dev/null | ms)
    echo "$B   $a   $b   $c   $d" >> $OUT
  done
  cat $OUT
fi
if [ "$1" = "m2" ] || [ -z "$1" ]; then
  for mode in "" "--graph"; do
    python bench.py --model DiffMa-XL/2 --use-mamba2 --batch-per-gpu 256 --cpu-steps 0 --steps 10 $mode > gpurun_out/bench_c4_xl2_mamba2_b256${mode:+_graph}.json 2>gpurun_out/m2.err || tail -3 gpurun_out/m2.err
    tail -1 gpurun_out/bench_c4_xl2_mamba2_b256${mode:+_graph}.json | ms
  done
  bash tools/pmc_cmd.sh m2_ssd "%ssd_bwd%|%ssd_fwd%" python $R/bench.py --model DiffMa-XL/2 --use-mamba2 --batch-per-gpu 256 --cpu-steps 0 --no-extras --steps 2 --warmup 1 --gemm-tuning frozen > /dev/null 2>&1
  cat gpurun_out/m2_ssd_pmc.txt | cut -c1-110
fi
if [ "$1" = "f32" ] || [ -z "$1" ]; then
  bash tools/pmc_cmd.sh scan_fwd_f32 "%scan_fwd_kernel<float%" python $R/tools/bench_kernels.py --batch 768 --dtype fp32 --only scan_fwd --iters 5 > /dev/null 2>&1
  cat gpurun_out/scan_fwd_f32_pmc.txt | cut -c1-110
fi
