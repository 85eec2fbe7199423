#!/bin/bash
# Timing-only ablation / probe builds of the software-pipelined attention kernel (ablation results are wrong by construction).
# usage: tools/attn_asm/build_pipe_variants.sh  -> tools/attn_asm/variants/libgrl_<name>.so ; run with tools/attn_asm/run_variants.py
set -e
ROOT=$(cd $(dirname $0)/../.. && pwd)
CS=$ROOT/grl_image_restoration_amd/csrc
OUT=$ROOT/tools/attn_asm/variants
mkdir -p $OUT
rm -f $OUT/libgrl_*.so
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=fast -Wno-unused-result -Wno-unused-value -Wno-inline-asm"
OBJS="linear.o linear_k576.o linear_k1152.o linear_split.o mlp.o qkv.o qkv_anchor.o attention.o attention_rows.o attention_bwd.o conv.o cab_conv2.o tail_regs.o misc.o grad.o"
build() {  # name, generator --abl, extra -D flags
  T=$(mktemp -d)
  python3 $ROOT/tools/attn_asm/gen_attn_pipe.py --out $T/pipe.inc --abl "$2"
  (cd $CS && /opt/rocm/bin/hipcc $FLAGS $3 -DPIPE_ASM_INC="\"$T/pipe.inc\"" -I$ROOT/include -I$CS -c attention_pipe.hip -o $T/ap.o)
  (cd $CS && /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $OBJS $T/ap.o -o $OUT/libgrl_$1.so)
  rm -rf $T
  echo built $1
}
if [ $# -gt 0 ]; then build "$@"; exit 0; fi
build a_base "" "" &
build b_nobias "nobias" "-DGRL_ABLATION" &
build c_nokv "nokv" "-DGRL_ABLATION" &
build d_nolds "nobias,nokv" "-DGRL_ABLATION" &
wait
build e_noexp "noexp" "-DGRL_ABLATION" &
build f_nomfma "nomfma" "-DGRL_ABLATION" &
build g_nodma "" "-DGRL_ABLATION -DPIPE_ABL_NODMA" &
build h_nobarrier_nodma "" "-DGRL_ABLATION -DPIPE_ABL_NODMA -DPIPE_ABL_NOBARRIER" &
wait
build i_valuonly "nobias,nokv,nomfma" "-DGRL_ABLATION -DPIPE_ABL_NODMA" &
build j_mfmaonly "nobias,nokv,noexp,nocvt" "-DGRL_ABLATION -DPIPE_ABL_NODMA" &
build z_debug "" "-DPIPE_DEBUG" &
wait
