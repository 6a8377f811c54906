"""The frequency-domain fine sync of the arithmetic contract against the plain definition (no GPU, no reference needed).

Since round 4 the oracle (and the kernel it is the twin of) never forms the time series of a frequency tweak or of the final grid:
scores and grid magnitudes come straight from the spectrum slice through an exact identity (oracle/ft8_oracle.c: fine_fscore,
fine_grid_freq).  This test restates the DEFINITION in numpy float64 -- tapered slice, 3200-point inverse FFT,
32-sample DFTs at clamped symbol positions (receiver.py:180-206) -- and requires the oracle's 79 x 8 grid to agree to 2e-6 of the
row maximum and its frequency tweak to be the definition's arg max (or within 1e-5 of it), for candidates all over the search window
including ones whose first or last symbols are read clamped."""
import numpy as np

import oracle as O
from pyft8_amd import synth

COSTAS = (3, 1, 4, 0, 6, 5, 2)
W6 = float(np.float32(-1.0 / 6.0))


def _series(spec, fb):
    step = (0.0 - np.pi) / 99.0
    y = np.array([0.0 if i == 99 else i * step + np.pi for i in range(100)])
    taper = 0.5 * (1.0 + np.cos(y))                                       # receiver.py:183-184
    X = np.zeros(3200, np.complex128)
    seg = spec[fb:fb + 850].astype(np.complex128); seg[750:] *= taper
    low = spec[fb - 150:fb].astype(np.complex128); low[:100] *= taper
    X[:850] = seg; X[3050:] = low
    return np.fft.ifft(X)


def _mags(z, tb, symbols):
    out = np.zeros((len(symbols), 8))
    n = np.arange(32)
    for i, s in enumerate(symbols):
        i0 = min(max(tb + 32 * s, 0), 3168)
        x = z[i0:i0 + 32]
        for t in range(8):
            out[i, t] = abs(np.sum(x * np.exp(-2j * np.pi * n * t / 32)))
    return out


def _score(z, tb):
    g = _mags(z, tb, range(36, 43))[:, :7]
    on = sum(g[a, COSTAS[a]] for a in range(7))
    return on + W6 * (g.sum() - on)


def _check(cfg_kwargs, frame, n_cands, want_clamped):
    audio = synth.make_frame(frame)
    cfg = O.default_config(**cfg_kwargs)
    grid = O.spectrogram(audio, cfg)
    spec = O.cycle_spectrum(audio, cfg)
    cands = O.sync_search(grid, cfg)
    assert len(cands) >= n_cands
    pick = np.linspace(0, len(cands) - 1, n_cands).astype(int)
    clamped = 0
    for ci in pick:
        c = cands[ci]
        r = O.fine(spec, c.f0_idx, c.h0_idx, cfg)
        tb0 = 8 * c.h0_idx + (1 if c.h0_idx < 0 else 0)
        tb, fb0 = tb0 + r["ttweak"], 50 * c.f0_idx
        z = _series(spec, fb0 + r["ftweak"])
        want = _mags(z, tb, range(79))
        got = r["sgrid"].astype(np.float64)
        assert np.abs(got - want).max() <= 2e-6 * want.max(), (c.f0_idx, c.h0_idx, np.abs(got - want).max() / want.max())
        clamped += int(tb < 0 or tb + 32 * 78 > 3168)
        # the frequency tweak: arg max of the definition's score over range(-32, 33, 8) at the chosen time tweak (first maximum)
        sc = [_score(_series(spec, fb0 + f), tb) for f in range(-32, 33, 8)]
        best = int(np.argmax(sc))
        chosen = (r["ftweak"] + 32) // 8
        assert chosen == best or abs(sc[chosen] - sc[best]) <= 1e-5 * abs(sc[best]), (c.f0_idx, c.h0_idx, r["ftweak"], sc)
    assert clamped >= want_clamped, clamped


def test_grid_and_frequency_tweak_equal_the_definition_default_window():
    _check({}, 4200, 24, 3)


def test_grid_and_frequency_tweak_equal_the_definition_wide_time_windows():
    _check(dict(h0_lo=-137, h0_hi=87, sync_score_min=70.0, max_cands=256), 4201, 16, 4)       # up to 34 symbols before sample 0
    _check(dict(h0_lo=-12, h0_hi=217, sync_score_min=70.0, max_cands=256), 4202, 16, 4)        # up to 33 symbols beyond sample 3168
