import sys, os, time
sys.path.insert(0, os.getcwd())
import baby_plonk_rust_amd as bp
from oracle import oracle as O
ctx = bp.Context(0)
h = ctx.srs_generate_progression(1000, 5, 3); ctx.srs_free(h)      # builds the generator table
for lg in (16, 20, 22):
    n = 1 << lg
    t0 = time.perf_counter(); h = ctx.srs_generate_progression(n, 0x1F2E3D4C5B6A7988, 0x10203); ctx.synchronize(); dt = time.perf_counter() - t0
    t0 = time.perf_counter(); h2 = ctx.srs_generate(n, 0x1234567); ctx.synchronize(); dt2 = time.perf_counter() - t0
    print("2^%d: progression %.2f ms, powers of tau %.2f ms" % (lg, 1e3 * dt, 1e3 * dt2), flush=True)
    if lg == 16:
        pts = ctx.srs_export(h)[:96 * 300]
        want = bytes(O.points_to_bytes96(O.points_progression(300, 0x1F2E3D4C5B6A7988, 0x10203)))
        print("first 300 points equal the oracle's:", pts == want)
    ctx.srs_free(h); ctx.srs_free(h2)
