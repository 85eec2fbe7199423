import sys, time, torch
sys.path.insert(0, "/root/repo")
from grl_image_restoration_amd import GRL, baseline_config
m = GRL(**baseline_config(3)).eval().cuda()
for B in (1, 2, 8):
    x = torch.rand(B, 3, 256, 256, device="cuda")
    for mode in ("eager", "graph"):
        m.enable_graph(mode == "graph")
        with torch.no_grad():
            for _ in range(3): m(x)
            torch.cuda.synchronize(); t0 = time.perf_counter()
            n = 5
            for _ in range(n): m(x)
            torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / n
        print(f"B={B} {mode}: {dt*1e3:.2f} ms/forward  {B*65536/dt/1e6:.3f} LQ-MP/s")
