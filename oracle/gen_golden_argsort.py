"""Golden vectors of the reference's third-party sort: np.argsort(float32) of THIS container's numpy (2.2.6, AVX512_SKX dispatch =
x86-simd-sort), which is what osd_012 orders its columns with (decoders.py:226).  Inputs are shaped like -abs(llr): tie-laden (AP bits
at 5.0), NaN-laden, all-NaN, special values, every length class of the library's network.  -> tests/golden/argsort_numpy.npz

    python oracle/gen_golden_argsort.py        (build container only; records numpy's version and CPU features)"""
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def vectors(rng):
    out = []
    for t in range(400):
        kind = t % 8
        n = 174 if t % 4 else int(rng.integers(2, 257))
        if kind < 2:
            a = -np.abs(rng.integers(0, rng.integers(2, 8), n)).astype(np.float32)
        elif kind == 2:
            a = -np.abs(np.round(rng.normal(0, 3, n), 1)).astype(np.float32)
        elif kind in (3, 4):
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
            a[rng.integers(0, n, 3)] = -np.inf
            a[rng.integers(0, n, 5)] = -0.0
        out.append(a)
    return out


def main():
    from numpy._core._multiarray_umath import __cpu_features__ as feat
    assert feat.get("AVX512_SKX"), "generate on the AVX-512 host the reference goldens were made on"
    vs = vectors(np.random.default_rng(20261003))
    x = np.full((len(vs), 256), np.float32(0), np.float32)
    order = np.full((len(vs), 256), -1, np.int32)
    n = np.array([len(v) for v in vs], np.int32)
    for i, v in enumerate(vs):
        x[i, :len(v)] = v
        order[i, :len(v)] = np.argsort(v)
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "argsort_numpy.npz"), x=x, n=n, order=order,
                        numpy=np.array(np.__version__), simd=np.array("AVX512_SKX"))
    print("wrote tests/golden/argsort_numpy.npz:", len(vs), "vectors")


if __name__ == "__main__":
    main()
