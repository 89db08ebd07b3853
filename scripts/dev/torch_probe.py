import time, sys
t0 = time.time()
def log(*a): print("[%.1fs]" % (time.time() - t0), *a, flush=True)
import torch
log("torch imported", torch.__version__, torch.cuda.is_available())
x = torch.zeros(4, device="cuda"); torch.cuda.synchronize(); log("alloc ok")
y = (x + 1).sum().item(); log("elementwise ok", y)
m = (x != 0).to(torch.uint8); torch.cuda.synchronize(); log("compare ok")
g = torch.Generator(device="cuda"); g.manual_seed(1); r = torch.randn(8, device="cuda", dtype=torch.float64, generator=g); torch.cuda.synchronize(); log("randn ok")
a = torch.randn(64, 1000, device="cuda", dtype=torch.float64); b = torch.randn(1000, 32, device="cuda", dtype=torch.float64)
c = a @ b; torch.cuda.synchronize(); log("dgemm ok")
sel = a[:, 0] > 0; d = a[sel]; torch.cuda.synchronize(); log("mask index ok")
