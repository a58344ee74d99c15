"""Which draws (seeds) of tests/test_gpu_y_full_size.py see the same ReLU pattern in the oracle and in the HIP path, in
both arithmetic modes?  Prints one line per (model, seed); the test's SEED table is filled from this.
    python tools/ubench/find_kink_free_draw.py [first_seed] [count]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "vae-time-series_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import test_gpu_y_full_size as tm  # noqa: E402

first = int(sys.argv[1]) if len(sys.argv) > 1 else 11
count = int(sys.argv[2]) if len(sys.argv) > 2 else 4
for kind in ("mixture", "gpd", "gaussian"):
    for seed in range(first, first + count):
        tm.SEED[kind] = seed
        tm._CACHE.pop(kind, None)
        res = []
        for mode in ("bf16x3", "f32"):
            try:
                tm.test_bench_sized_step_matches_oracle(kind, mode)
                res.append(f"{mode}: ok")
            except AssertionError as e:
                res.append(f"{mode}: {str(e)[:160]}")
        print(kind, seed, " | ".join(res), flush=True)
        if all(r.endswith("ok") for r in res):
            break
