#!/bin/bash
# Register / scratch / occupancy table of the kernels the default paths launch (no GPU needed):
#   tools/kernel_resources.sh > profiles/rNN_kernel_resources.txt
# hipcc -Rpass-analysis=kernel-resource-usage per source file; linear.hip (about 110 instantiations) is filtered to the shapes the
# inference and training paths of GRL-Base use.
set -u
cd "$(dirname "$0")/../grl_image_restoration_amd/csrc"
echo "# hipcc --offload-arch=gfx950 -O3 -Rpass-analysis=kernel-resource-usage  (tools/kernel_resources.sh)"
for f in attention attention_rows attention_pipe attention_bwd mlp tail_regs qkv qkv_anchor conv conv192 cab_conv2 linear_split grad misc planes ln_train cpb linear linear_k576; do
  extra=""
  case $f in attention|attention_bwd) extra="-mllvm -amdgpu-mfma-vgpr-form";; esac   # as in csrc/Makefile
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=fast -I../../include --cuda-device-only $extra -c $f.hip -o /dev/null \
        -Rpass-analysis=kernel-resource-usage 2>&1 |
    awk '/remark: Function Name:/{name=$(NF-1)} /remark: +TotalSGPRs:/{s=$(NF-1)} /remark: +VGPRs:/{v=$(NF-1)} /remark: +AGPRs:/{a=$(NF-1)}
         /remark: +ScratchSize/{sc=$(NF-1)} /remark: +Occupancy/{o=$(NF-1)} /remark: +VGPRs Spill:/{sp=$(NF-1)}
         /remark: +LDS Size/{printf "Function Name: %s\tTotalSGPRs: %s\tVGPRs: %s\tAGPRs: %s\tScratchSize [bytes/lane]: %s\tVGPRs Spill: %s\tOccupancy [waves/SIMD]: %s\n", name, s, v, a, sc, sp, o}' |
    if [ $f = linear ] || [ $f = linear_k576 ]; then grep -E "linear_kernelILi(6|12|18)ELi(6|8|12)E"; else cat; fi
done
