#!/bin/bash
# timing-only ablation builds of one translation unit:  build_variants_generic.sh <file.hip> name1:-DFLAG1 name2:-DFLAG2 ...
set -e
ROOT=$(cd $(dirname $0)/../.. && pwd)
CS=$ROOT/grl_image_restoration_amd/csrc
OUT=$ROOT/tools/attn_asm/variants
mkdir -p $OUT
SRC=$1; shift
OBJ=${SRC%.hip}.o
FLAGS="-DGRL_ABLATION --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=fast -Wno-unused-result -Wno-unused-value -Wno-inline-asm"
ALL="linear.o linear_k576.o linear_k1152.o mlp.o qkv.o qkv_anchor.o attention.o attention_rows.o attention_bwd.o conv.o cab_conv2.o tail_regs.o misc.o grad.o"
REST=$(echo $ALL | sed "s/\b$OBJ\b//")
for spec in "$@"; do
  name=${spec%%:*}; defs=${spec#*:}
  ( T=$(mktemp -d); cd $CS && /opt/rocm/bin/hipcc $FLAGS $defs -I$ROOT/include -c $SRC -o $T/v.o && /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $REST $T/v.o -o $OUT/libgrl_$name.so && rm -rf $T && echo built $name ) &
done
wait
