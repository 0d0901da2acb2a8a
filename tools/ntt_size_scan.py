import sys, os, time
sys.path.insert(0, os.getcwd())
import torch
import baby_plonk_rust_amd as bp
ctx = bp.Context(0)
for lg in range(10, 27):
    n = 1 << lg
    v = torch.empty(n * 4, dtype=torch.int64, device="cuda")
    ctx.synthetic_scalars_device(v.data_ptr(), n, 77 + lg)
    for _ in range(3): ctx.ntt_device(v.data_ptr(), lg)
    best = 1e9
    for _ in range(5):
        ctx.ntt_device(v.data_ptr(), lg)
        best = min(best, ctx.ntt_stats()["device_ms"])
    st = ctx.ntt_stats()
    print("ntt 2^%d: %.4f ms  passes %d  %.3f ns/element  %.3f ns per element-stage" % (lg, best, st["passes"], best * 1e6 / n, best * 1e6 / n / lg), flush=True)
    del v
