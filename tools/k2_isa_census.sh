#!/bin/bash
# instruction census of one K2 instantiation (default: the DiffMa mixer's bf16 call pattern): static counts over the whole kernel,
# i.e. one 8-step chunk body + prologue / epilogue.   tools/k2_isa_census.sh [object] [mangled-name regex]
CS=$(dirname "$0")/../diffma-diffusion-mamba_amd/csrc
OBJ=${1:-$CS/scan_bwd_bf16.o}
PAT=${2:-'scan_bwd_kernelINS_6bf16_tES1_Li16ELi1ELb0ELb1ELi2ELb0E'}
TMP=$(mktemp -d)
BIN=/opt/rocm/lib/llvm/bin
$BIN/llvm-objcopy --dump-section .hip_fatbin=$TMP/fat.bin $OBJ
$BIN/clang-offload-bundler --unbundle --type=o --input=$TMP/fat.bin --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --output=$TMP/dev.co
$BIN/llvm-objdump -d --no-show-raw-insn $TMP/dev.co | awk -v pat="$PAT" '
  /^[0-9a-f]+ <.*>:$/ { on = ($0 ~ pat) }
  on && /^[ \t]+[a-z]/ { print $1 }' > $TMP/ops.txt
echo "total instructions: $(wc -l < $TMP/ops.txt)"
python3 - "$TMP/ops.txt" <<'PY'
import sys, collections
ops = [l.strip() for l in open(sys.argv[1])]
c = collections.Counter(ops)
grp = collections.Counter()
for o, n in c.items():
    if o.startswith("v_exp") or o.startswith("v_log") or o.startswith("v_rcp") or o.startswith("v_rsq"): grp["transcendental"] += n
    elif o.startswith("v_pk_"): grp["packed"] += n
    elif o.startswith("v_mfma"): grp["mfma"] += n
    elif o.startswith("v_"): grp["valu other"] += n
    elif o.startswith("ds_"): grp["lds"] += n
    elif o.startswith("buffer_") or o.startswith("global_") or o.startswith("scratch_"): grp["vmem"] += n
    elif o.startswith("s_waitcnt") or o.startswith("s_nop"): grp[o] += n
    elif o.startswith("s_"): grp["salu"] += n
    else: grp["other"] += n
print(dict(grp))
print("top:", c.most_common(28))
PY
rm -rf $TMP
