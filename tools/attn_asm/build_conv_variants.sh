#!/bin/bash
# Timing-only ablation builds of csrc/conv192.hip (results wrong by construction) -> tools/attn_asm/variants/libgrl_<name>.so
set -e
ROOT=$(cd $(dirname $0)/../.. && pwd)
CS=$ROOT/grl_image_restoration_amd/csrc
OUT=$ROOT/tools/attn_asm/variants
mkdir -p $OUT; rm -f $OUT/libgrl_*.so
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=fast -Wno-unused-result -Wno-unused-value -Wno-inline-asm"
OBJS="linear.o linear_k576.o linear_k1152.o linear_split.o mlp.o qkv.o qkv_anchor.o attention.o attention_rows.o attention_pipe.o attention_bwd.o conv.o cab_conv2.o tail_regs.o misc.o grad.o"
build() {
  T=$(mktemp -d)
  (cd $CS && /opt/rocm/bin/hipcc $FLAGS $2 -I$ROOT/include -I$CS -c conv192.hip -o $T/c.o && /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $OBJS $T/c.o -o $OUT/libgrl_$1.so)
  rm -rf $T; echo built $1
}
build a_base "" &
build b_nomfma "-DGRL_ABLATION -DC9_ABL_NOMFMA" &
build c_nostore "-DGRL_ABLATION -DC9_ABL_NOSTORE" &
build d_noresid "-DGRL_ABLATION -DC9_ABL_NORESID" &
wait
build e_noepi "-DGRL_ABLATION -DC9_ABL_NOSTORE -DC9_ABL_NORESID" &
build f_noinput "-DGRL_ABLATION -DC9_ABL_NOINPUT" &
build g_nodma "-DGRL_ABLATION -DC9_ABL_NODMA" &
build h_mfmaonly "-DGRL_ABLATION -DC9_ABL_NOSTORE -DC9_ABL_NORESID -DC9_ABL_NOINPUT -DC9_ABL_NODMA" &
wait
