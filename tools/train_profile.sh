#!/bin/bash
# rocprofv3 kernel statistics of the training leg (BASELINE configs[4] shape): gpurun_out/prof/train_kernel_stats.txt
set -u
ROOT=$(pwd); export TMPDIR=/tmp; OUT=$ROOT/gpurun_out/prof; mkdir -p $OUT
cd /tmp; rm -rf /tmp/prof_tr
cat > /tmp/train_steps.py <<PY
import sys, torch
sys.path.insert(0, "$ROOT")
from grl_image_restoration_amd import GRL, FusedAdamW, baseline_config
m = GRL(**baseline_config(5)).cuda().train()
opt = FusedAdamW(m.parameters(), lr=2e-4, weight_decay=1e-4)
g = torch.Generator().manual_seed(0)
lq, gt = torch.rand(8, 3, 64, 64, generator=g).cuda(), torch.rand(8, 3, 256, 256, generator=g).cuda()
for _ in range(3):
    opt.zero_grad(set_to_none=True); loss = (m(lq) - gt).abs().mean(); loss.backward(); opt.step()
torch.cuda.synchronize()
PY
rocprofv3 --kernel-trace --stats -d /tmp/prof_tr -o r -- python /tmp/train_steps.py > /tmp/train_prof.log 2>&1
DB=$(find /tmp/prof_tr -name "*.db" | head -1)
echo "# rocprofv3 --kernel-trace --stats -- 3 training steps of GRL-Base x4 SR, batch 8 x 64x64 LQ (autograd over the HIP kernels + FusedAdamW)" > $OUT/train_kernel_stats.txt
python $ROOT/tools/rocprof_summary.py "$DB" 40 >> $OUT/train_kernel_stats.txt 2>&1
cd $ROOT
