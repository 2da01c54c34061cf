#!/bin/bash
# register / LDS / spill budget of the kernels of one translation unit (from the code-object metadata notes of its .o):
#   tools/kernel_regs.sh <unit, e.g. scan_bwd_bf16> [regex on the mangled kernel name]
CS=$(dirname "$0")/../diffma-diffusion-mamba_amd/csrc
OBJ=$CS/${1:-scan_bwd_bf16}.o
[ -f "$1" ] && OBJ=$1
PAT=${2:-.}
TMP=$(mktemp -d)
BIN=/opt/rocm/lib/llvm/bin
$BIN/llvm-objcopy --dump-section .hip_fatbin=$TMP/fat.bin $OBJ
$BIN/clang-offload-bundler --unbundle --type=o --input=$TMP/fat.bin --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --output=$TMP/dev.co
$BIN/llvm-readelf --notes $TMP/dev.co | PAT="$PAT" python3 -c "
import sys, re, os, subprocess
txt = sys.stdin.read()
pat = os.environ['PAT']
rows = []
for blk in txt.split('.agpr_count:')[1:]:
    g = lambda k: (re.search(r'\.%s:\s+(\S+)' % k, blk) or [None, '?'])[1]
    name = g('name')
    if re.search(pat, name):
        rows.append(('vgpr %3s agpr %3s sgpr %3s lds %6s spill_v %3s scratch %5s  ' % (g('vgpr_count'), blk.split()[0], g('sgpr_count'), g('group_segment_fixed_size'), g('vgpr_spill_count'), g('private_segment_fixed_size')), name))
dem = subprocess.run(['c++filt'] + [n for _, n in rows], capture_output=True, text=True).stdout.splitlines() if rows else []
for (a, _), d in sorted(zip(rows, dem), key=lambda t: t[1]):
    print(a + d[:170])
"
rm -rf $TMP
