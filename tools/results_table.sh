run() { timeout -k 5 400 python bench.py --cpu-steps 0 "$@" 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$*', '| ms/step', d['ms_per_step'], '| value', d['value'], '| roof', d['roofline']['kernel'], d['roofline']['frac'])"; }
run --mode sample --graph --model DiffMa-B/4 --batch-per-gpu 8 --dtype fp32 --steps 50 --warmup 5
run --mode sample --graph --model DiffMa-B/4 --batch-per-gpu 64 --dtype fp32 --steps 50 --warmup 5
run --mode sample --graph --model DiffMa-B/4 --batch-per-gpu 64 --dtype bf16 --steps 50 --warmup 5
run --model DiffMa-L/2 --batch-per-gpu 256
run --model DiffMa-L/2 --batch-per-gpu 512
run --model DiffMa-XL/2 --use-mamba2 --batch-per-gpu 64 --steps 5 --warmup 2
run --model DiffMa-XL/2 --batch-per-gpu 64 --steps 5 --warmup 2
run --mode sample --graph --sampler ddim50 --model DiffMa-XXL/2 --batch-per-gpu 8 --steps 30 --warmup 5
run --mode sample --graph --sampler ddim50 --model DiffMa-XXL/2 --batch-per-gpu 64 --steps 30 --warmup 5
run --model DiffMa-L/2 --batch-per-gpu 8 --steps 20
run --model DiffMa-L/2 --batch-per-gpu 8 --steps 20 --graph
run --mode sample --graph --model DiffMa-L/2 --batch-per-gpu 1 --steps 50 --warmup 5
run --mode sample --graph --model DiffMa-L/2 --batch-per-gpu 8 --steps 50 --warmup 5
run --mode sample --graph --model DiffMa-L/2 --batch-per-gpu 64 --steps 50 --warmup 5
