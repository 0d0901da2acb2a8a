#!/usr/bin/env python3
"""Randomised differential run: the C ABI against closed forms and the CPU oracle over random sizes, window widths,
table settings, scalar distributions, NTT lengths and polynomial operators.  Not part of the pytest suite (its fixed
cases are a subset of what this explores); run it on a GPU box with a time budget:
    python tools/fuzz_parity.py --seconds 240 --seed 1
Exits non-zero with the failing case printed."""
import argparse
import os
os.environ.setdefault("BABY_PLONK_LIBRARY", "exp")        # this tool sets BP_* knobs: only the experiment build reads them (make -C baby_plonk_rust_amd/csrc exp)
import random
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

import baby_plonk_rust_amd as bp
from oracle import oracle as O
from tests import bigint_model as M
from tests.gpu_common import Q, oracle_dot

ap = argparse.ArgumentParser()
ap.add_argument("--seconds", type=float, default=120)
ap.add_argument("--seed", type=int, default=1)
args = ap.parse_args()
rnd = random.Random(args.seed)
ctx = bp.default_context()
group = bp.Context([0, 0, 0])          # the same library calls through a three-shard context (bp_init_multi) on this one card
group4 = bp.Context([0, 0, 0, 0])      # four members: one host NTT is cut over them (BP_NTT_GROUP_SPLIT_FROM lowered below)
os.environ["BP_NTT_GROUP_SPLIT_FROM"] = "11"
t_end = time.time() + args.seconds
counts = {"msm": 0, "ntt": 0, "poly": 0}


def scalars(n, kind):
    if kind == "random":
        return O.splitmix_scalars(n, rnd.getrandbits(60))
    if kind == "small":
        return bp.scalars_from_ints([rnd.randrange(1 << rnd.choice([1, 4, 16, 40])) for _ in range(n)])
    if kind == "equal":
        return np.repeat(bp.scalar_from_int(rnd.randrange(Q))[None, :], n, axis=0)
    if kind == "sparse":
        v = [0] * n
        for _ in range(max(1, n // 50)):
            v[rnd.randrange(n)] = rnd.randrange(Q)
        return bp.scalars_from_ints(v)
    if kind == "edges":
        return bp.scalars_from_ints([rnd.choice([0, 1, Q - 1, Q - 2, 2**254, 2**255 % Q, (1 << 128) - 1]) for _ in range(n)])
    raise ValueError(kind)


t_note = time.time() + 60
while time.time() < t_end:
    if time.time() > t_note:                 # a line a minute: a silent GPU command is taken to be hung after seven
        print("...", counts, flush=True)
        t_note = time.time() + 60
    which = rnd.choice(["msm", "msm", "ntt", "poly"])
    if which == "msm":
        n = rnd.choice([rnd.randrange(1, 40), rnd.randrange(1, 3000), rnd.randrange(1, 70000), rnd.randrange(60000, 300000)])
        a, d = rnd.randrange(Q), rnd.randrange(Q)
        srs_len = n + rnd.choice([0, 0, 3, 100])
        c = ctx if rnd.random() < 0.7 else group
        ctx_single, ctx = ctx, c
        h = ctx.srs_generate_progression(srs_len, a, d)
        mode = rnd.choice(["plain", "tables", "tables"])
        c_plain = rnd.choice([None, 4, 7, 10, 13, 16])
        if mode == "tables":        # widths above 16 take the partitioned sort (only when 8 n >= 2^width, else the plain path answers)
            # 256 + w: tables of every bit position, width-w NAF digits (256 rows: kept to short SRS here)
            naf_ok = srs_len <= 40000 and bp._lib.EXPERIMENT          # every-position tables exist in the experiment build only
            ctx.srs_precompute(h, rnd.choice([0, 0, 4, 6, 9, 12, 15, 16, 17, 18, 19, 20] + ([256 + 6, 256 + 9, 256 + 13, 256 + 16, 256 + 19] if naf_ok else [])))
        os.environ.pop("BP_MSM_C", None)
        if c_plain:
            os.environ["BP_MSM_C"] = str(c_plain)
        if rnd.random() < 0.6:
            os.environ["BP_MSM_CHUNK"] = str(rnd.choice([4, 5, 8, 13, 16, 32, 61, 64]))
        kind = rnd.choice(["random", "small", "equal", "sparse", "edges"])
        sc = scalars(n, kind)
        first = rnd.randrange(0, srs_len - n + 1) if rnd.random() < 0.3 else 0
        got = bp.sum_partials(ctx.msm_partial(h, sc, first=first))
        want = M.enc96(M.ec_mul(oracle_dot(sc, (a + first * d) % Q, d)))
        if got != want:
            print("MSM MISMATCH", dict(n=n, srs_len=srs_len, mode=mode, c_plain=c_plain, chunk=os.environ.get("BP_MSM_CHUNK"), kind=kind, first=first,
                                        table=ctx.srs_table_info(h), a=a, d=d, seed=args.seed, shards=ctx.n_shards()))
            sys.exit(1)
        ctx.srs_free(h)
        ctx = ctx_single
        os.environ.pop("BP_MSM_C", None)
        os.environ.pop("BP_MSM_CHUNK", None)
    elif which == "ntt":
        k = rnd.randrange(0, 19)
        x = scalars(1 << k, rnd.choice(["random", "small", "edges"]))
        inv = rnd.random() < 0.5
        nctx = group4 if rnd.random() < 0.3 else ctx
        got = nctx.ntt(x, inverse=inv)
        if not (got == O.ntt_fast(x, inverse=inv)).all():
            print("NTT MISMATCH", dict(k=k, inverse=inv, seed=args.seed, members=nctx.ntt_stats()["members"]))
            sys.exit(1)
        batch = rnd.randrange(1, 5)
        if k <= 14:
            xs = np.concatenate([scalars(1 << k, "random") for _ in range(batch)])
            gb = ctx.ntt_batch(xs.reshape(batch, 1 << k, 4), inverse=inv)
            for j in range(batch):
                if not (np.asarray(gb).reshape(batch, 1 << k, 4)[j] == O.ntt_fast(xs.reshape(batch, 1 << k, 4)[j], inverse=inv)).all():
                    print("NTT BATCH MISMATCH", dict(k=k, batch=batch, j=j, seed=args.seed))
                    sys.exit(1)
    else:
        na, nb = rnd.randrange(1, 3000), rnd.randrange(1, 300)
        A, B = scalars(na, rnd.choice(["random", "small", "sparse"])), scalars(nb, "random")
        P = lambda v: bp.Polynomial(v, bp.BASIS_MONOMIAL, ctx)
        prod = (P(A) * P(B)).values
        if not (prod == O.poly_binop("poly_mul_fast", A, B)).all():
            print("POLY MUL MISMATCH", dict(na=na, nb=nb, seed=args.seed))
            sys.exit(1)
        div_kind = rnd.choice(["general", "linear", "binomial"])
        if div_kind == "linear":
            D = bp.scalars_from_ints([rnd.randrange(Q), 1])
        elif div_kind == "binomial":
            m = rnd.randrange(1, 64)
            D = bp.scalars_from_ints([rnd.randrange(1, Q)] + [0] * (m - 1) + [rnd.randrange(1, Q)])
        else:
            D = B
        num = O.poly_binop("poly_mul_fast", A, D) if rnd.random() < 0.7 else A      # exact or with remainder
        got = (P(num) / P(D)).values
        want = O.poly_binop("poly_div", num, D)
        if got.shape != want.shape or not (got == want).all():
            print("POLY DIV MISMATCH", dict(na=na, nb=len(D), kind=div_kind, seed=args.seed))
            sys.exit(1)
        x = bp.scalar_from_int(rnd.randrange(Q))
        if not (P(A).coeffs_evaluate(x) == O.poly_eval(A, x, fast=True)).all():
            print("POLY EVAL MISMATCH", dict(na=na, seed=args.seed))
            sys.exit(1)
    counts[which] += 1
print("fuzz ok:", counts, "seed", args.seed)
