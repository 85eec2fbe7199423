#!/bin/bash
# Timing-only ablation builds of the row-streaming attention kernel (results are wrong by construction).
# usage: tools/attn_asm/build_variants.sh  -> tools/attn_asm/variants/libgrl_<name>.so ; run with tools/attn_asm/run_variants.py
set -e
ROOT=$(cd $(dirname $0)/../.. && pwd)
CS=$ROOT/grl_image_restoration_amd/csrc
OUT=$ROOT/tools/attn_asm/variants
mkdir -p $OUT
FLAGS="-DGRL_ABLATION --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=fast -Wno-unused-result -Wno-unused-value -Wno-inline-asm"
OBJS="linear.o linear_k576.o linear_k1152.o mlp.o qkv.o qkv_anchor.o attention.o attention_bwd.o conv.o cab_conv2.o tail_regs.o misc.o grad.o"
build() {  # name, generator --abl, extra -D flags
  T=$(mktemp -d)
  cp $CS/attention_rows.hip $CS/common.h $CS/attn_common.h $CS/grl_hip_internal.h $T/
  python3 $ROOT/tools/attn_asm/gen_attn_loop.py --out $T/attn_rows_asm.inc --abl "$2"
  (cd $T && /opt/rocm/bin/hipcc $FLAGS $3 -I$ROOT/include -I$CS -c attention_rows.hip -o ar.o)
  (cd $CS && /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $OBJS $T/ar.o -o $OUT/libgrl_$1.so)
  rm -rf $T
  echo built $1
}
build base "" "" &
build prio "prio" "" &
build priopv "priopv" "" &
build stagger "" "-DROWS_STAGGER" &
wait
