"""bench.py reads its profiler-derived constants (PMC traffic, the static instruction mix, the register-only rates) from the committed files of
the newest round under profiles/ -- never from typed-in numbers.  This checks the lookup on the CPU: the files exist, parse, are
the current round's, and carry plausible values; a renamed or reformatted profile must fail here, not silently fall back to an old round."""
import importlib.util
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench():
    spec = importlib.util.spec_from_file_location("bench_module", os.path.join(ROOT, "bench.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_register_only_rates_come_from_the_newest_round():
    b = _bench()
    newest = b.PROFILE_ROUNDS[0]
    assert b.UBENCH["source"] == "profiles/%s_ubench_valu_floor.txt" % newest, b.UBENCH
    assert 6.5e9 < b.BARE_ADDS_PER_S < 8.5e9                       # g1_add_mixed28 on registers only, whole chip
    assert 1200 < b.UBENCH["butterfly_clk"] < 1500                 # fr29_butterfly, clocks per 64 butterflies per SIMD
    assert 1.0e11 < b.BARE_BUTTERFLIES_PER_S < 1.4e11


def test_traffic_and_instruction_mix_lookups():
    b = _bench()
    newest = b.PROFILE_ROUNDS[0]
    t, name = b.profile_lookup("hbm_traffic.json", "msm_accumulate_2p20_c20_tables")
    assert name == newest + "_hbm_traffic.json" and t["algorithmic_bytes_per_launch"] == 128 << 20
    assert t["hbm_bytes_per_launch"] > t["algorithmic_bytes_per_launch"] and newest in t["source"]
    n, name = b.profile_lookup("hbm_traffic.json", "ntt_2p24")
    assert name == newest + "_hbm_traffic.json" and n["algorithmic_bytes_per_launch"] == 64 << 24
    mix, name = b.profile_lookup("msm_accumulate_instr_mix.json", "msm_accumulate<2>")
    assert name == newest + "_msm_accumulate_instr_mix.json" and mix["v_mad_u64_u32"] > 3000 and mix["quarter_rate"] + mix["full_rate"] == mix["valu"]
    assert b.profile_lookup("hbm_traffic.json", "no such key") == (None, None)
    hbm, issue = b.msm_roofline(1 << 20, 1.96e-3, 13631451, 20, True, "msm_accumulate_2p20_c20_tables")
    assert abs(hbm["frac"] - 128 * (1 << 20) / 1.96e-3 / 8e12) < 1e-9 and hbm["designed_frac"] > 10 * hbm["frac"]
    assert hbm["traffic_source"].startswith("profiles/" + newest) and 0.8 < issue["frac"] < 1.05 and 0.8 < issue["bare_kernel_peak"]["frac"] < 1.0
