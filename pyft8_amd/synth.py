"""Synthetic FT8 frame generator (host/numpy): the workload source for tests and bench.

This is the build's own generator for BASELINE.json configs 1-4;
it models what the reference transmitter does (reference: PyFT8/transmitter.py:41-70
GFSK BT=2.0 waveform, :97-124 standard-message packing, :181-223 CRC-14 / LDPC(174,91)
encode / Gray map / Costas framing) but is written from the FT8 protocol definition,
not translated.  Known-answer check (tests/test_synth.py): "CQ G1OJS IO90" must give
the 79-tone string printed by the reference's self-test.

Recipe for one frame (config 1): 50 signals, standard i3=1 messages with plausible
callsigns, f0 ~ U[200,2800] Hz, start = 0.5 s + U[-0.5,+1.0] s, SNR ~ U[lo,hi] dB in
2500 Hz (WSJT-X convention), unit-variance white noise scaled to sigma = 1000 counts,
clipped to int16.  RNG = numpy Philox keyed on (seed_base + frame index).
"""
import numpy as np

from .ft8_tables import GEN_HEX

FS = 12000
NSPS = 1920                    # samples per symbol at 12 kHz (6.25 baud)
NFRAME = 180000
COSTAS = (3, 1, 4, 0, 6, 5, 2)
GRAY = (0, 1, 3, 2, 5, 6, 4, 7)
SEED_BASE = 0x46543800

_A0 = " 0123456789ABCDEFGHIJKLMNOPQRSTUVWXYZ"
_A1 = "0123456789ABCDEFGHIJKLMNOPQRSTUVWXYZ"
_A2 = "0123456789"
_A3 = " ABCDEFGHIJKLMNOPQRSTUVWXYZ"
NTOKENS, MAX22 = 2063592, 4194304

_GEN = [int(h, 16) >> 1 for h in GEN_HEX]   # 83 rows x 91 bits


# ----------------------------------------------------------------------------- packing
def pack_c28(call):
    """Standard callsign or token -> 28-bit integer (FT8 protocol, i3=1)."""
    if call in ("DE", "QRZ", "CQ"):
        return ("DE", "QRZ", "CQ").index(call)
    c = call
    if len(c) < 3 or len(c) > 6:
        raise ValueError(f"not a standard call: {call!r}")
    if not c[2].isdigit():
        c = " " + c
    c = (c + "      ")[:6]
    if not c[2].isdigit():
        raise ValueError(f"not a standard call: {call!r}")
    n = _A0.index(c[0])
    n = n * 36 + _A1.index(c[1])
    n = n * 10 + _A2.index(c[2])
    n = n * 27 + _A3.index(c[3])
    n = n * 27 + _A3.index(c[4])
    n = n * 27 + _A3.index(c[5])
    return n + NTOKENS + MAX22


def pack_g15(txt):
    """Grid / report / RRR / RR73 / 73 -> (g15, ir)."""
    if txt == "RRR":
        return 32402, 0
    if txt == "RR73":
        return 32403, 0
    if txt == "73":
        return 32404, 0
    ir = 0
    t = txt
    if t.startswith("R") and len(t) > 1 and t[1] in "+-":
        ir, t = 1, t[1:]
    if t[0] in "+-":
        return 32400 + 35 + int(t), ir
    if len(txt) == 4:
        v = (ord(txt[0]) - 65) * 18 + (ord(txt[1]) - 65)
        return (v * 10 + int(txt[2])) * 10 + int(txt[3]), 0
    raise ValueError(txt)


def pack77(call_a, call_b, extra):
    """Standard message (i3=1, or 2 if a '/P' is present) -> 77-bit int."""
    def split(c):
        if c.endswith("/P") or c.endswith("/R"):
            return c[:-2], 1
        return c, 0
    a, pa = split(call_a)
    b, pb = split(call_b)
    i3 = 2 if (call_a.endswith("/P") or call_b.endswith("/P")) else 1
    g15, ir = pack_g15(extra)
    v = pack_c28(a)
    v = (v << 1) | pa
    v = (v << 28) | pack_c28(b)
    v = (v << 1) | pb
    v = (v << 1) | ir
    v = (v << 15) | g15
    v = (v << 3) | i3
    return v


def crc14(bits77):
    """CRC-14 (poly 0x2757) of the 77-bit message zero-extended to 82 bits."""
    r = 0
    for i in range(96):
        b = (bits77 >> (76 - i)) & 1 if i < 77 else 0
        top = (r >> 13) & 1
        r = ((r << 1) & 0x3FFF) | b
        if top:
            r ^= 0x2757
    return r


def encode174(bits77):
    """77 bits -> 174-bit LDPC codeword (int, MSB = codeword bit 0)."""
    m91 = (bits77 << 14) | crc14(bits77)
    par = 0
    for row in _GEN:
        par = (par << 1) | (bin(m91 & row).count("1") & 1)
    return (m91 << 83) | par


def tones79(bits77):
    cw = encode174(bits77)
    syms = [GRAY[(cw >> (171 - 3 * i)) & 7] for i in range(58)]
    return list(COSTAS) + syms[:29] + list(COSTAS) + syms[29:] + list(COSTAS)


# ----------------------------------------------------------------------------- waveform
def _gfsk_pulse(bt=2.0):
    from math import erf, log, pi, sqrt
    k = pi * sqrt(2.0 / log(2.0)) * bt
    t = (np.arange(3 * NSPS) - 1.5 * NSPS) / NSPS
    return np.array([0.5 * (erf(k * (x + 0.5)) - erf(k * (x - 0.5))) for x in t])


_PULSE = None


def tones_to_phase(tones, f0):
    """79 tones -> the GFSK phase (radians) at each of the 79*1920 samples.  Modelled on the reference transmitter's
    symbols_to_complex_audio (transmitter.py:52-70): same pulse, same dummy edge symbols; the reference accumulates the phase
    inclusively, so its waveform is this one advanced by exactly one sample (tests/test_synth.py pins that against a golden of
    the reference's own output)."""
    global _PULSE
    if _PULSE is None:
        _PULSE = _gfsk_pulse()
    n = len(tones)
    ext = [tones[0]] + list(tones) + [tones[-1]]          # dummy edge symbols
    dphi = np.zeros((n + 2) * NSPS + 2 * NSPS)
    for i, t in enumerate(ext):
        dphi[i * NSPS:i * NSPS + 3 * NSPS] += t * _PULSE
    dphi = dphi[int(1.5 * NSPS) + NSPS // 2: int(1.5 * NSPS) + NSPS // 2 + n * NSPS]
    dphi = 2 * np.pi * (f0 + 6.25 * dphi) / FS
    return np.cumsum(dphi) - dphi


def tones_to_wave(tones, f0):
    """79 tones -> real GFSK waveform, 79*1920 samples, unit amplitude."""
    global _PULSE
    if _PULSE is None:
        _PULSE = _gfsk_pulse()
    n = len(tones)
    ext = [tones[0]] + list(tones) + [tones[-1]]          # dummy edge symbols
    dphi = np.zeros((n + 2) * NSPS + 2 * NSPS)
    for i, t in enumerate(ext):
        dphi[i * NSPS:i * NSPS + 3 * NSPS] += t * _PULSE
    dphi = dphi[int(1.5 * NSPS) + NSPS // 2: int(1.5 * NSPS) + NSPS // 2 + n * NSPS]
    dphi = 2 * np.pi * (f0 + 6.25 * dphi) / FS
    phi = np.cumsum(dphi) - dphi
    w = np.sin(phi)
    nr = NSPS // 8
    ramp = 0.5 * (1 - np.cos(np.pi * np.arange(nr) / nr))
    w[:nr] *= ramp
    w[-nr:] *= ramp[::-1]
    return w


# ----------------------------------------------------------------------------- messages
_PFX1 = "ACDEHJLOPSTUVXYZ"       # one-letter prefixes valid as <L><digit> without the BFGIKMNRW 3rd-digit trap
_PFX2 = ["DL", "EA", "ON", "OK", "OZ", "PA", "SP", "SV", "UA", "UR", "YO", "YU", "HB", "HA", "LZ", "LY",
         "ES", "OH", "SM", "LA", "EI", "GM", "GW", "IK", "IZ", "JA", "VK", "ZL", "VE", "WA", "KB", "KC"]


def random_call(rng):
    L = "ABCDEFGHIJKLMNOPQRSTUVWXYZ"
    if rng.random() < 0.35:
        p = _PFX1[rng.integers(len(_PFX1))]
        # keep first char a valid 1-char prefix, 2nd a digit
        call = p + str(rng.integers(10))
        nsuf = rng.integers(2, 4)
    else:
        call = _PFX2[rng.integers(len(_PFX2))] + str(rng.integers(10))
        nsuf = rng.integers(1, 4)
    call += "".join(L[rng.integers(26)] for _ in range(nsuf))
    return call


def random_message(rng):
    a, b = random_call(rng), random_call(rng)
    k = rng.integers(4)
    L = "ABCDEFGHIJKLMNOPQR"
    if k == 0:
        grid = L[rng.integers(18)] + L[rng.integers(18)] + str(rng.integers(10)) + str(rng.integers(10))
        return ("CQ", b, grid)
    if k == 1:
        return (a, b, f"{int(rng.integers(-24, 20)):+03d}")
    if k == 2:
        return (a, b, "R" + f"{int(rng.integers(-24, 20)):+03d}")
    return (a, b, ("RR73", "73", "RRR")[rng.integers(3)])


def make_frame(index, n_signals=50, snr_range=(-10.0, 10.0), seed_base=SEED_BASE, return_truth=False, freq_range=(200.0, 2800.0)):
    """One synthetic 15-s frame -> int16[180000] (and the truth list if asked).  freq_range: where the carriers go (wide-range tests)."""
    rng = np.random.Generator(np.random.Philox(key=seed_base + int(index)))
    x = rng.standard_normal(NFRAME)
    truth = []
    for _ in range(n_signals):
        msg = random_message(rng)
        f0 = rng.uniform(*freq_range)
        t0 = 0.5 + rng.uniform(-0.5, 1.0)
        snr = rng.uniform(*snr_range)
        amp = np.sqrt(2.0 * (2500.0 / 6000.0) * 10.0 ** (snr / 10.0))
        w = tones_to_wave(tones79(pack77(*msg)), f0)
        i0 = int(round(t0 * FS))
        n = min(len(w), NFRAME - i0)
        x[i0:i0 + n] += amp * w[:n]
        truth.append(dict(msg=" ".join(msg), f0=float(f0), t0=float(t0), snr=float(snr)))
    y = np.clip(np.rint(x * 1000.0), -32768, 32767).astype(np.int16)
    return (y, truth) if return_truth else y


def frame_from_words(index, words77, snr_range=(0.0, 8.0), seed_base=SEED_BASE):
    """A frame carrying the given 77-bit words (any message type the transmitter could send: i3 = 4 non-standard calls, hashed
    calls, ...) evenly spread over 300..2700 Hz -- for parity tests of the message layer inside the whole pipeline."""
    rng = np.random.Generator(np.random.Philox(key=seed_base + 77000000 + int(index)))
    x = rng.standard_normal(NFRAME)
    n = len(words77)
    for k, w77 in enumerate(words77):
        f0 = 300.0 + 2400.0 * (k + 0.5) / n + rng.uniform(-3.0, 3.0)
        t0 = 0.5 + rng.uniform(-0.3, 0.8)
        amp = np.sqrt(2.0 * (2500.0 / 6000.0) * 10.0 ** (rng.uniform(*snr_range) / 10.0))
        w = tones_to_wave(tones79(int(w77)), f0)
        i0 = int(round(t0 * FS))
        x[i0:i0 + len(w)] += amp * w
    return np.clip(np.rint(x * 1000.0), -32768, 32767).astype(np.int16)


def make_batch(start, count, **kw):
    return np.stack([make_frame(start + i, **kw) for i in range(count)])


# ----------------------------------------------------------------------------- device generator support
# Record layout of csrc/ft8rx.hip:SynthSig (one per signal).
SIGNAL_DTYPE = np.dtype([("f0", "<f8"), ("cum", "<f8", (82,)), ("amp", "<f4"), ("i0", "<i4"), ("ext", "u1", (84,))], align=True)


def pulse_cumsum():
    """Q[k] = sum_{t<k} pulse[t], k = 0..5760 (the integrated GFSK frequency pulse)."""
    global _PULSE
    if _PULSE is None:
        _PULSE = _gfsk_pulse()
    return np.concatenate([[0.0], np.cumsum(_PULSE)])


def _tones79_batch(words):
    """tones79 for many 77-bit words at once (numpy GF(2) encode; same tones as tones79, which the tests pin to the reference)."""
    n = len(words)
    crc = np.array([crc14(w) for w in words], np.uint64)
    m91 = np.zeros((n, 91), np.uint8)
    w = np.array([[(x >> (76 - i)) & 1 for i in range(77)] for x in words], np.uint8).reshape(n, 77)
    m91[:, :77] = w
    m91[:, 77:] = (crc[:, None] >> (13 - np.arange(14, dtype=np.uint64))[None, :]) & np.uint64(1)
    gen = np.array([[(row >> (90 - i)) & 1 for i in range(91)] for row in _GEN], np.uint8)       # [83][91]
    par = (m91.astype(np.int32) @ gen.T.astype(np.int32)) & 1
    cw = np.concatenate([m91, par.astype(np.uint8)], axis=1)                                    # [n][174]
    sym = (cw[:, 0::3] << 2) | (cw[:, 1::3] << 1) | cw[:, 2::3]
    g = np.asarray(GRAY, np.uint8)[sym]
    c = np.broadcast_to(np.asarray(COSTAS, np.uint8), (n, 7))
    return np.concatenate([c, g[:, :29], c, g[:, 29:], c], axis=1)


def _signal_table_chunk(args):
    start, count, n_signals, snr_range, seed_base = args
    Q = pulse_cumsum()
    qs = np.zeros(81)
    qs[0], qs[1] = Q[3840], Q[1920]
    recs = np.zeros((count, max(1, n_signals)), SIGNAL_DTYPE)
    truth, words = [], []
    for fi in range(count):
        rng = np.random.Generator(np.random.Philox(key=seed_base + int(start + fi)))
        tr = []
        for s in range(n_signals):
            msg = random_message(rng)
            f0 = rng.uniform(200.0, 2800.0)
            t0 = 0.5 + rng.uniform(-0.5, 1.0)
            snr = rng.uniform(*snr_range)
            words.append(pack77(*msg))
            r = recs[fi, s]
            r["f0"] = f0
            r["amp"] = np.sqrt(2.0 * (2500.0 / 6000.0) * 10.0 ** (snr / 10.0))
            r["i0"] = int(round(t0 * FS))
            tr.append(dict(msg=" ".join(msg), f0=float(f0), t0=float(t0), snr=float(snr)))
        truth.append(tr)
    if n_signals and count:
        tones = _tones79_batch(words)
        ext = np.concatenate([tones[:, :1], tones, tones[:, -1:]], axis=1)                       # [n][81]
        cum = np.cumsum(ext.astype(np.float64) * (Q[5760] - qs)[None, :], axis=1)
        recs["cum"][:, :n_signals, 1:] = cum.reshape(count, n_signals, 81)
        recs["ext"][:, :n_signals, :81] = ext.reshape(count, n_signals, 81)
    return recs, truth


def device_signal_table(start, count, n_signals=50, snr_range=(-10.0, 10.0), seed_base=SEED_BASE, workers=None):
    """Signal parameters for ft8rx_synth_frames: same recipe as make_frame (messages, f0, t0, SNR drawn from the
    per-frame Philox stream); the noise itself is generated on the device.  -> (records[count, n_signals], truth).
    Large tables (the 8192-frame shards of config 3: 400 k messages to pack and encode) are built by `workers` child processes
    (default: one per allowed CPU up to 32 for >= 512 frames) -- plain `python -m pyft8_amd.synth` children, numpy only, so
    neither a GPU-initialised parent is forked nor its __main__ re-imported."""
    import os
    if workers is None:
        ncpu = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
        workers = min(32, ncpu) if count >= 512 else 1
    workers = max(1, min(int(workers), count // 64 or 1))
    if workers == 1:
        return _signal_table_chunk((start, count, n_signals, tuple(snr_range), seed_base))
    import pickle
    import subprocess
    import sys
    import tempfile
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    per = (count + workers - 1) // workers
    jobs = [(start + o, min(per, count - o)) for o in range(0, count, per)]
    env = dict(os.environ, PYTHONPATH=root + os.pathsep + os.environ.get("PYTHONPATH", ""))
    with tempfile.TemporaryDirectory() as d:
        procs = [subprocess.Popen([sys.executable, "-m", "pyft8_amd.synth", os.path.join(d, f"{i}.pkl"), str(s0), str(n), str(n_signals),
                                   repr(float(snr_range[0])), repr(float(snr_range[1])), str(seed_base)], cwd=root, env=env)
                 for i, (s0, n) in enumerate(jobs)]
        if any(p.wait() != 0 for p in procs):
            raise RuntimeError("device_signal_table: a table worker failed")
        parts = [pickle.load(open(os.path.join(d, f"{i}.pkl"), "rb")) for i in range(len(jobs))]
    return np.concatenate([p[0] for p in parts]), [t for p in parts for t in p[1]]


# ----------------------------------------------------------------------------- numpy twins of the device generator (k_synth)
DEVICE_SEED = 0x4654385F53594E54        # default seed of _lib.Handle.synth_frames


def _philox4x32_10(c0, c1, c2, c3, k0, k1):
    """Philox4x32-10 on uint32 arrays (the counter-based generator of csrc/kernels/synth.hpp)."""
    M0, M1 = np.uint64(0xD2511F53), np.uint64(0xCD9E8D57)
    c0, c1, c2, c3 = (np.asarray(x, np.uint32).copy() for x in (c0, c1, c2, c3))
    k0, k1 = np.uint32(k0), np.uint32(k1)
    for _ in range(10):
        p0 = M0 * c0.astype(np.uint64)
        p1 = M1 * c2.astype(np.uint64)
        n0 = (p1 >> np.uint64(32)).astype(np.uint32) ^ c1 ^ k0
        n1 = p1.astype(np.uint32)
        n2 = (p0 >> np.uint64(32)).astype(np.uint32) ^ c3 ^ k1
        n3 = p0.astype(np.uint32)
        c0, c1, c2, c3 = n0, n1, n2, n3
        k0 = np.uint32((int(k0) + 0x9E3779B9) & 0xFFFFFFFF)
        k1 = np.uint32((int(k1) + 0xBB67AE85) & 0xFFFFFFFF)
    return c0, c1, c2, c3


def device_noise(frame_index, seed=DEVICE_SEED):
    """The unit-variance noise k_synth adds to frame `frame_index` (float64[180000]): counter = (sample group, frame index), key =
    seed; two Box-Muller pairs per group of four samples."""
    g = np.arange(NFRAME // 4, dtype=np.uint32)
    z = np.zeros_like(g)
    r = _philox4x32_10(g, np.full_like(g, np.uint32(frame_index & 0xFFFFFFFF)), z, z, seed & 0xFFFFFFFF, seed >> 32)
    u = [(x.astype(np.float64) + 0.5) * (1.0 / 4294967296.0) for x in r]
    ra, rb = np.sqrt(-2.0 * np.log(u[0])), np.sqrt(-2.0 * np.log(u[2]))
    out = np.empty(NFRAME)
    out[0::4] = ra * np.cos(6.283185307179586 * u[1])
    out[1::4] = ra * np.sin(6.283185307179586 * u[1])
    out[2::4] = rb * np.cos(6.283185307179586 * u[3])
    out[3::4] = rb * np.sin(6.283185307179586 * u[3])
    return out


def device_frame(frame_index, table_row, n_signals, noise=True, seed=DEVICE_SEED):
    """numpy twin of one k_synth frame from its signal table row (device_signal_table): int16[180000]."""
    x = device_noise(frame_index, seed) if noise else np.zeros(NFRAME)
    for s in range(n_signals):
        r = table_row[s]
        tones = [int(t) for t in r["ext"][1:80]]
        w = tones_to_wave(tones, float(r["f0"]))
        i0 = int(r["i0"])
        lo, hi = max(i0, 0), min(i0 + len(w), NFRAME)
        x[lo:hi] += float(r["amp"]) * w[lo - i0:hi - i0]
    return np.clip(np.rint(x * 1000.0), -32768, 32767).astype(np.int16)


if __name__ == "__main__":          # table worker of device_signal_table: out.pkl start count n_signals snr_lo snr_hi seed_base
    import pickle
    import sys
    _a = sys.argv[1:]
    _res = _signal_table_chunk((int(_a[1]), int(_a[2]), int(_a[3]), (float(_a[4]), float(_a[5])), int(_a[6])))
    with open(_a[0], "wb") as _f:
        pickle.dump(_res, _f, protocol=pickle.HIGHEST_PROTOCOL)
