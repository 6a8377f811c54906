"""np.argsort(float32) of the INSTALLED numpy (the reference's third-party sort, decoders.py:226) against the oracle's restatement
(oracle/ft8_oracle.c: ft8o_argsort_f32 -- x86-simd-sort's AVX-512 argsort network for n <= 256, libstdc++ std::sort for vectors with NaN).

    python tools/argsort_check.py [n_vectors = 1000000] [seed = 0]    ->  one summary line per kind; exit code 1 on any mismatch

Vector kinds (two thirds of the vectors have the OSD length 174, the rest every length 2..256):
  ties       -|k|, k from 2..7 small integers (few distinct keys)        rounded    -|N(0,3)| rounded to 0.1 (many small tie groups)
  ap         -|N(0,4)| with 30 % of the entries at -5.0 (the AP mask)     nan-mix    tie-laden keys with a random fraction of NaN
  all-nan    all NaN (a NaN-poisoned BP output), sometimes one number     specials   random keys with -inf and -0.0 entries
Meaningful only where the installed numpy dispatches argsort to AVX512_SKX (np.show_runtime()); prints that first."""
import ctypes as C
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))


def main():
    import oracle as O
    n_vec = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
    L = O.lib()
    L.ft8o_argsort_f32.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
    from numpy._core._multiarray_umath import __cpu_features__ as feat
    print(f"numpy {np.__version__}, AVX512_SKX {'yes' if feat.get('AVX512_SKX') else 'NO (np.argsort is a different algorithm here)'}")
    names = ["ties", "ties", "ties", "rounded", "ap", "nan-mix", "all-nan", "specials"]
    tot, bad = {}, {}
    t0 = time.time()
    out = np.zeros(256, np.int32)
    for t in range(n_vec):
        kind = t % 8
        n = 174 if t % 3 else int(rng.integers(2, 257))
        if kind < 3:
            a = -np.abs(rng.integers(0, rng.integers(2, 8), n)).astype(np.float32)
        elif kind == 3:
            a = -np.abs(np.round(rng.normal(0, 3, n), 1)).astype(np.float32)
        elif kind == 4:
            a = -np.abs(rng.normal(0, 4, n)).astype(np.float32)
            a[rng.random(n) < 0.3] = -5.0
        elif kind == 5:
            a = -np.abs(rng.integers(0, 4, n)).astype(np.float32)
            a[rng.random(n) < rng.random()] = np.nan
        elif kind == 6:
            a = np.full(n, np.nan, np.float32)
            if t % 16 == 6:
                a[rng.integers(0, n)] = -1.0
        else:
            a = -np.abs(rng.normal(0, 4, n)).astype(np.float32)
            if rng.random() < 0.3:
                a[rng.integers(0, n, 3)] = -np.inf
            if rng.random() < 0.3:
                a[rng.integers(0, n, 5)] = -0.0
        assert L.ft8o_argsort_f32(a.ctypes.data, n, out.ctypes.data) == 0
        k = names[kind]
        tot[k] = tot.get(k, 0) + 1
        if not np.array_equal(np.argsort(a), out[:n]):
            bad[k] = bad.get(k, 0) + 1
    for k in sorted(tot):
        print(f"  {k:9s} {tot[k]:9d} vectors, {bad.get(k, 0)} differ from np.argsort")
    nb = sum(bad.values())
    print(f"{n_vec} vectors in {time.time() - t0:.0f} s: {nb} mismatches")
    return 1 if nb else 0


if __name__ == "__main__":
    sys.exit(main())
