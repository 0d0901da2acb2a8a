import sys, os, random
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
import baby_plonk_rust_amd as bp
from baby_plonk_rust_amd.synthetic import chained_multiplications, Q
ctx = bp.default_context()
def used():
    torch.cuda.synchronize()
    free, total = torch.cuda.mem_get_info()
    return (total - free) / 2**20
n = 1 << 14
cols, pk = chained_multiplications(n, 3)
base = None
for cycle in range(6):
    setup = bp.Setup.generate_srs(n + 6, 12345 + cycle, ctx)
    circuit = bp.Circuit(pk, ctx)
    prover = bp.Prover(setup, circuit)
    for _ in range(5):
        blob = prover.prove_with_blinding(cols[0], cols[1], cols[2], None, list(range(1, 12)))
    h = ctx.srs_generate_progression(50000, 3, 5)
    ctx.srs_precompute(h, 0)
    sc = bp.scalars_from_ints([random.randrange(Q) for _ in range(1000)])
    for _ in range(20):
        ctx.msm(h, sc)
    ctx.srs_free(h)
    circuit.free()
    ctx.srs_free(setup.handle)
    u = used()
    if cycle == 1:
        base = u
    print("cycle", cycle, "device MiB in use: %.1f" % u, flush=True)
assert abs(used() - base) < 8, "device memory grows across load/free cycles"
print("no growth after warm-up")

# group contexts come and go (members, worker threads, coset shares and their work contexts), and the one-call seam's workspaces
from oracle import oracle as O
base = None
img = None
for cycle in range(5):
    many = bp.Context([0, 0, 0, 0])
    setup = bp.Setup.generate_srs(n + 6, 777 + cycle, many)
    circuit = bp.Circuit(pk, many)
    prover = bp.Prover(setup, circuit)
    for _ in range(3):
        blob2 = prover.prove_with_blinding(cols[0], cols[1], cols[2], None, list(range(1, 12)))
    m = (1 << 17) + 3
    h = many.srs_generate_progression(m, 3, 5)
    if img is None:
        img = many.srs_export_projective144(h)
    sc = O.splitmix_scalars(m, 5)
    assert many.msm_projective144(img, sc) == ctx.msm_projective144(img, sc)
    many.close()
    u = used()
    if cycle == 1:
        base = u
    print("group cycle", cycle, "device MiB in use: %.1f" % u, flush=True)
assert abs(used() - base) < 8, "device memory grows across group-context cycles"
print("no growth across group contexts")
