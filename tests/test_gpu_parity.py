"""GPU parity tests (run with -m gpu on an MI355X): every call goes through the C ABI of libft8rx.so
(pyft8_amd._lib.Handle) and is compared with the CPU oracle on the same inputs -- bit-exact, floats
included, because both sides implement the same explicitly ordered IEEE arithmetic -- and
with the golden vectors captured from the reference (tolerance 1e-4 where floats are involved)."""
import json
import os

import numpy as np
import pytest

import oracle as O
from conftest import GOLDEN_FRAMES, load_golden

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def H():
    from pyft8_amd import _lib
    h = _lib.Handle(max_frames=16)
    yield h
    h.close()


@pytest.fixture(scope="module")
def ocfg():
    from pyft8_amd import _lib
    return O.default_config(**_lib.fft_plans())


def bits_equal(a, b):
    """Bitwise equality of float32 arrays; NaNs must sit at the same positions (their sign/payload bits are
    not compared: x86 generates 0xFFC00000 for 0/0, gfx950 0x7FC00000)."""
    a, b = np.ascontiguousarray(a, np.float32), np.ascontiguousarray(b, np.float32)
    na, nb = np.isnan(a), np.isnan(b)
    return np.array_equal(na, nb) and np.array_equal(a.view(np.uint32)[~na], b.view(np.uint32)[~nb])


def test_native_library_loaded():
    from pyft8_amd import _lib
    assert _lib.lib() is not None and _lib.lib().ft8rx_device_count() >= 1


def test_elementary_functions_bit_exact(H):
    rng = np.random.default_rng(0)
    x = np.concatenate([np.logspace(-38, 38, 200001), rng.uniform(0.5, 2.0, 200000), [0.0, 1.0, 1e-45, 3.4e38],
                        np.nextafter(np.float32(1.0), np.float32([0.0, 2.0])), np.float32(2.0) ** np.arange(-126, 128)]).astype(np.float32)
    assert bits_equal(H.math_probe(0, x), O.log10f(x))
    # (the kernel's divisions run without the compiler's range repairs, ft8_div_inrange: the tanh quotient P / Q has operands in
    # [4.9e-3, 0.91] whatever x is -- tiny, denormal, huge and special arguments included here)
    t = np.concatenate([rng.uniform(-12, 12, 400000), rng.standard_normal(20000) * 1e-3, np.logspace(-44, 1, 4001), -np.logspace(-44, 1, 4001),
                        [0.0, -0.0, 1.0, -1.0, 50.0, -50.0, 1e-45, -1e-45, 1.2e-38, 3e38, -3e38, 7.90531111, np.nextafter(np.float32(7.90531111), np.float32(8)),
                         np.nan, np.inf, -np.inf]]).astype(np.float32)
    got, want = H.math_probe(1, t), O.tanhf(t)
    assert bits_equal(got, want)
    z = t == 0
    assert np.array_equal(np.signbit(got[z]), np.signbit(t[z]))          # tanh(-0.0) = -0.0


def test_fft_bit_exact(H, ocfg):
    rng = np.random.default_rng(1)
    from pyft8_amd import _lib
    plans = _lib.fft_plans()
    for n, key in [(1920, "plan1920"), (3200, "plan3200"), (300, "plan300"), (320, "plan320")]:
        x = (rng.standard_normal(n) + 1j * rng.standard_normal(n)).astype(np.complex64) * 1000
        got = H.math_probe(2, x)
        want = O.fft(x, plans[key])
        assert bits_equal(got.view(np.float32), want.view(np.float32)), n
        ref = np.fft.fft(x.astype(np.complex128))
        assert np.abs(got - ref).max() / np.abs(ref).max() < 2e-6


def test_spectrogram_bit_exact(H, ocfg):
    audio = np.stack([load_golden(n)[0] for n in GOLDEN_FRAMES])
    g = H.spectrogram(audio)
    for i, name in enumerate(GOLDEN_FRAMES):
        want = O.spectrogram(audio[i], ocfg)
        assert bits_equal(g[i], want), name
        gold = load_golden(name)[1]
        ref, got = gold["grid_rows"], g[i][gold["grid_rows_idx"]]
        strong = ref > ref.max(axis=1, keepdims=True) - 60.0
        assert np.abs(got - ref)[strong].max() < 1e-3


def test_sync_search_exact(H, ocfg):
    for name in GOLDEN_FRAMES:
        audio, gold, js = load_golden(name)
        grid = O.spectrogram(audio, ocfg)
        f0, h0, sc, cnt = H.sync_search(grid)
        want = O.sync_search(grid, ocfg)
        n = int(cnt[0])
        assert n == len(want) == js["n_cands"]
        assert list(f0[0, :n]) == [c.f0_idx for c in want] == list(gold["f0_idx"])
        assert list(h0[0, :n]) == [c.h0_idx for c in want] == list(gold["h0_idx"])
        assert bits_equal(sc[0, :n], np.array([c.score for c in want], np.float32))


def test_llr_grid_exact(H, ocfg):
    for name in GOLDEN_FRAMES:
        audio, gold, js = load_golden(name)
        grid = O.spectrogram(audio, ocfg)
        n = js["n_cands"]
        llr, sd, snr = H.llr_grid(grid, np.zeros(n, np.int32), gold["f0_idx"], gold["h0_idx"])
        for i in range(n):
            w_llr, w_sd, w_snr, ok = O.db_to_llr(O.payload(grid, gold["f0_idx"][i], gold["h0_idx"][i]))
            assert bits_equal(llr[i], w_llr) and np.float32(sd[i]) == np.float32(w_sd) and snr[i] == w_snr == gold["grid_snr"][i]


def test_cycle_spectrum_bit_exact(H, ocfg):
    audio = np.stack([load_golden(n)[0] for n in GOLDEN_FRAMES[:3]])
    s = H.cycle_spectrum(audio)
    for i, name in enumerate(GOLDEN_FRAMES[:3]):
        want = O.cycle_spectrum(audio[i], ocfg)
        assert bits_equal(s[i].view(np.float32), want.view(np.float32)), name
        gold = load_golden(name)[1]
        keep = gold["spec_idx"] < O.SPEC_BINS
        ref = gold["spec_val"][keep]
        assert np.abs(s[i][gold["spec_idx"][keep]] - ref).max() / np.sqrt(np.mean(np.abs(ref) ** 2)) < 1e-5


def test_fine_sync_exact(H, ocfg):
    for name in GOLDEN_FRAMES:
        audio, gold, js = load_golden(name)
        spec = O.cycle_spectrum(audio, ocfg)
        idx = gold["fine_idx"]
        if len(idx) == 0:
            continue
        f0, h0 = gold["f0_idx"][idx], gold["h0_idx"][idx]
        r = H.fine(spec, np.zeros(len(idx), np.int32), f0, h0, want_sgrid=True)
        assert list(r["ttweak"]) == list(gold["fine_tt"]) and list(r["ftweak"]) == list(gold["fine_ft"])
        assert list(r["nsync"]) == list(gold["fine_nsync"])
        for k in range(len(idx)):
            w = O.fine(spec, f0[k], h0[k], ocfg)
            assert (r["ret"][k], r["ttweak"][k], r["ftweak"][k], r["nsync"][k]) == (w["ret"], w["ttweak"], w["ftweak"], w["nsync"])
            assert bits_equal(r["sgrid"][k], w["sgrid"])
            if w["ret"] != 0:
                assert bits_equal(r["llr"][k], w["llr"]) and np.float32(r["sd"][k]) == np.float32(w["sd"]) and r["snr"][k] == w["snr"]


def test_fine_sync_clamp_rows_are_bit_identical(H, ocfg):
    """receiver.py:189-195 reads symbol s at z[clip(tb + 32 s, 0, 3168) : +32]: every symbol that starts AT or beyond sample 3168 (or at
    or before 0) reads the same 32 samples, so its grid row is bit-identical to the others' -- the exact |LLR| ties osd_012's argsort
    then sees.  Stage entry ft8rx_fine on late / early candidates of a synthetic frame: GPU == oracle bit for bit, the clamped rows
    (the boundary symbol included, round 6) are one and the same row, and the test set really holds boundary cases."""
    from pyft8_amd import synth
    audio = synth.make_frame(61000, n_signals=50, snr_range=(-10.0, 10.0))
    spec = O.cycle_spectrum(audio, ocfg)
    f0 = np.arange(100, 900, 8, dtype=np.int32)
    trip = [(f, h) for h in (84, 88, 92, 96, 100, 120, 150, 180, 200, 216, -20, -60, -100, -136) for f in f0[:12]] + [(int(f), 88 + 4 * (i % 8)) for i, f in enumerate(f0)]
    # ... and far-out candidates (k_fine_td: the time-domain form with clamped reads, one series per tweak)
    trip += [(int(f), h) for h in (221, 230, 252, 300, 372, 400, 577, -141, -150, -168, -200, -312, -600, -898) for f in f0[:6]]
    f0s, h0s = np.array([t[0] for t in trip], np.int32), np.array([t[1] for t in trip], np.int32)
    r = H.fine(spec, np.zeros(len(trip), np.int32), f0s, h0s, want_sgrid=True)
    n_boundary = n_clamped = 0
    for k, (f, h) in enumerate(trip):
        w = O.fine(spec, f, h, ocfg)
        assert (r["ret"][k], r["ttweak"][k], r["ftweak"][k], r["nsync"][k]) == (w["ret"], w["ttweak"], w["ftweak"], w["nsync"]), (f, h)
        assert bits_equal(r["sgrid"][k], w["sgrid"]), (f, h)
        tb = 8 * h + (1 if h < 0 else 0) + int(w["ttweak"])
        starts = tb + 32 * np.arange(79)
        hi = np.nonzero(starts >= 3168)[0]
        lo = np.nonzero(starts <= 0)[0]
        g = r["sgrid"][k].reshape(79, 8)
        for grp, strict in ((hi, (starts > 3168).any()), (lo, (starts < 0).any())):
            if len(grp) >= 2 and strict:
                n_clamped += 1
                assert all(g[i].tobytes() == g[grp[0]].tobytes() for i in grp), (f, h, tb, grp[:3])
                n_boundary += int((starts[grp] == 3168).any() or (starts[grp] == 0).any())
    assert n_clamped > 40 and n_boundary >= 3, (n_clamped, n_boundary)


def test_ldpc_exact(H):
    for name in GOLDEN_FRAMES:
        audio, gold, js = load_golden(name)
        for nc0, its in sorted(set(zip(gold["bp_nc0max"].tolist(), gold["bp_iters"].tolist()))):
            sel = np.where((gold["bp_nc0max"] == nc0) & (gold["bp_iters"] == its))[0]
            ok, lo, hi, nits, has, out = H.ldpc(gold["bp_llr_in"][sel], nc0, its)
            for j, k in enumerate(sel):
                w_ok, w_bits, w_nits, w_out = O.ldpc(gold["bp_llr_in"][k], nc0, its)
                assert bool(ok[j]) == w_ok == (js["bp_result"][k] is not None)
                if w_ok:
                    assert ((int(hi[j]) << 64) | int(lo[j])) == w_bits and nits[j] == w_nits == gold["bp_nits"][k]
                else:
                    assert bool(has[j]) == (w_out is not None) == bool(gold["bp_has_out"][k])
                    if w_out is not None:
                        assert bits_equal(out[j], w_out)          # NaN payloads included


def test_ldpc_extension_knobs_match_oracle(H):
    """30 iterations (BASELINE config 2) has no reference counterpart; the oracle is the checker."""
    audio, gold, js = load_golden("synth_100000")
    x = gold["bp_llr_in"]
    ok, lo, hi, nits, has, out = H.ldpc(x, 90, 30)
    for k in range(len(x)):
        w_ok, w_bits, w_nits, w_out = O.ldpc(x[k], 90, 30)
        assert bool(ok[k]) == w_ok and (not w_ok or (((int(hi[k]) << 64) | int(lo[k])) == w_bits and nits[k] == w_nits))
        if w_out is not None:
            assert bits_equal(out[k], w_out)


def test_osd_exact(H):
    for name in GOLDEN_FRAMES:
        audio, gold, js = load_golden(name)
        x = gold["osd_llr_in"]
        # (62, 2) / (63, 5) straddle the one-word / two-word flip storage of the kernel, (91, 91) is every trial the reference's
        # signature allows (decoders.py:244-272: all 91 basis positions singly, then all 4095 pairs)
        for s, d in [(30, 2), (40, 3), (0, 0)] + ([(62, 2), (63, 5), (91, 3), (91, 91)] if name == GOLDEN_FRAMES[0] else []):
            ok, lo, hi, trial = H.osd(x, s, d)
            for k in range(len(x)):
                w_ok, w_bits, w_trial, _ = O.osd(x[k], s, d)
                assert bool(ok[k]) == w_ok, (name, k, s, d)
                if w_ok:
                    assert ((int(hi[k]) << 64) | int(lo[k])) == w_bits and trial[k] == w_trial
                if (s, d) == (30, 2):
                    assert w_ok == (js["osd_result"][k] is not None)


def test_osd_ties_and_nans(H):
    """The column order of osd_012 is np.argsort(-abs(llr)) of the reference's numpy, equal keys and NaNs included (oracle:
    ft8o_argsort_f32, pinned to np.argsort): AP-style exact ties, many equal magnitudes, vectors with some / only NaN (the library's
    std::sort path: k_osd_nan), zeros, infinities -- outcome, word and trial index equal the oracle's."""
    rng = np.random.default_rng(5)
    x = rng.standard_normal((96, 174)).astype(np.float32) * 3
    x[:, :29] = np.where(rng.random((96, 29)) < 0.5, 5.0, -5.0)        # AP-style exact ties
    x[10:20, 40:60] = np.nan
    x[20:24] = np.nan
    x[24:28] = 0.0
    x[64:80] = np.round(x[64:80])                                       # few distinct magnitudes: ties everywhere
    x[80:84, 100:110] = np.inf
    x[84:88, ::3] = np.nan
    x[88:96] = np.where(rng.random((8, 174)) < 0.5, 5.0, -5.0)         # every key equal
    for s, d in [(30, 2), (91, 4)]:
        ok, lo, hi, trial = H.osd(x, s, d)
        for k in range(len(x)):
            w_ok, w_bits, w_trial, _ = O.osd(x[k], s, d)
            assert bool(ok[k]) == w_ok and (not w_ok or (((int(hi[k]) << 64) | int(lo[k])) == w_bits and trial[k] == w_trial))


def test_osd_order3_and_distance_gate_exact(H):
    """Extension knobs (BASELINE config 4 "OSD depth-3"; no reference counterpart): triple flips over the least reliable basis
    positions and the Hamming-distance acceptance gate, GPU vs the oracle run with the same knobs -- outcome, word, trial index
    and the accepted codeword's distance."""
    rng = np.random.default_rng(11)
    x = np.concatenate([load_golden(n)[1]["osd_llr_in"] for n in GOLDEN_FRAMES[:3]])[:96]
    noisy = (rng.standard_normal((64, 174)) * 2.5).astype(np.float32)          # pure noise: whatever decodes is a false decode
    x = np.concatenate([x, noisy])
    hits = 0
    for s, d, t, g in [(30, 2, 10, 0), (30, 2, 30, 0), (30, 2, 30, 30), (30, 2, 0, 28), (40, 3, 36, 40), (0, 0, 12, 0)]:
        ok, lo, hi, trial, hd = H.osd(x, s, d, t, g, want_hd=True)
        for k in range(len(x)):
            w_ok, w_bits, w_trial, _, w_hd = O.osd(x[k], s, d, t, g, want_hd=True)
            assert bool(ok[k]) == w_ok, (k, s, d, t, g)
            if w_ok:
                hits += 1
                assert ((int(hi[k]) << 64) | int(lo[k])) == w_bits and trial[k] == w_trial and hd[k] == w_hd, (k, s, d, t, g)
                assert g == 0 or hd[k] <= g
    assert hits > 20


def test_crc_and_validity_exact(H):
    from pyft8_amd import synth
    rng = np.random.default_rng(2)
    words = []
    for k in range(4000):
        b = ((int(rng.integers(0, 2 ** 63)) << 14) ^ int(rng.integers(0, 2 ** 63))) & ((1 << 77) - 1)
        if k % 3 == 0:
            b = (b & ~7) | 1
        if k % 5 == 0:
            b = (b & ~7) | 4
        if k % 7 == 0:
            b = synth.pack77(*synth.random_message(rng))
        words.append(b)
    words += [0, 1, 6257895 << 49 | 1 << 3 | 1]
    v = H.valid77(words)
    assert [bool(x) for x in v] == [O.valid77(b) for b in words]
    cw = []
    for b in words[:600]:
        m91 = (b << 14) | synth.crc14(b)
        if len(cw) % 4 == 3:
            m91 ^= 1 << int(rng.integers(0, 91))
        cw.append([1.0 if (m91 >> (90 - k)) & 1 else -1.0 for k in range(91)])
    cw = np.array(cw, np.float32)
    res, lo, hi = H.crc_valid(cw)
    for k in range(len(cw)):
        r, bits = O.crc_valid91(cw[k])
        assert res[k] == r and (r == 0 or ((int(hi[k]) << 64) | int(lo[k])) == bits)


def _check_frame(rec, cnt, ev, evc, audio, js, ocfg):
    from pyft8_amd import messages as M
    r = O.decode_frame(audio, ocfg)
    n = int(cnt)
    assert n == len(r["cands"])
    for i, c in enumerate(r["cands"]):
        g = rec[i]
        assert (g["f0_idx"], g["h0_idx"]) == (c.f0_idx, c.h0_idx)
        assert np.float32(g["score"]) == np.float32(c.score)
        assert int(g["status"]) == c.status, (i, int(g["status"]), c.status)
        assert np.float32(g["grid_sd"]) == np.float32(c.grid_sd) and g["snr_grid"] == c.snr_grid
        if c.status in (1, 4, 5) and (c.status != 1 or c.ipass >= 2):
            assert (g["ttweak"], g["ftweak"], g["nsync"]) == (c.ttweak, c.ftweak, c.nsync)
            assert np.float32(g["fine_sd"]) == np.float32(c.fine_sd) and g["snr_fine"] == c.snr_fine
        if c.status == 1:
            assert (int(g["ipass"]), int(g["ap"]), int(g["method"]), int(g["n_its"])) == (c.ipass, c.ap, c.method, c.n_its), i
            assert (int(g["msg_lo"]), int(g["msg_hi"])) == (c.msg_lo, c.msg_hi)
    msgs = M.package_frame(rec, n, ev, int(evc), cyclestart_string="700101_000015")
    o_txt = [" ".join(m["msg_tuple"]) for m in r["msgs"]]
    g_txt = [" ".join(m["msg_tuple"]) for m in msgs]
    assert g_txt == o_txt
    if js is not None:
        assert g_txt == [" ".join(m["msg_tuple"]) for m in js["messages"]]
        for m, ref in zip(msgs, js["messages"]):
            for key in ("tsec", "fHz", "their_snr", "all_txt_format", "tweaks", "decode_notes", "cyclestart_string", "their_tx_cycle"):
                assert m[key] == ref[key], (key, m[key], ref[key])
            assert list(m["msg_tuple"]) == ref["msg_tuple"]
    return len(g_txt)


def test_decode_batch_golden_frames(H, ocfg):
    """The whole hot path on the reference's own fixtures: identical message strings, order and dict fields."""
    audio = np.stack([load_golden(n)[0] for n in GOLDEN_FRAMES])
    rec, cnt, ev, evc = H.decode_batch(audio)
    for i, name in enumerate(GOLDEN_FRAMES):
        _check_frame(rec[i], cnt[i], ev[i], evc[i], audio[i], load_golden(name)[2], ocfg)


def test_light_goldens_on_the_gpu():
    """33 more frames pinned to the REAL reference (tests/golden/light_frames.json, oracle/gen_golden_light.py; round 6 added two
    wide-time-window frames whose OSD outcome hangs on the clamp-boundary symbol, one at max_cands = 400 on the deep layouts and two with
    a search_time_range far beyond the fine-sync series): every Receiver kwargs
    set of the live cross-check, the four frames whose OSD outcome hangs on how np.argsort orders equal keys (plain goldens since round
    5: the kernel sorts as the reference's numpy does) and the one frame where the build knowingly deviates (last-ulp LLR) with its
    expected-difference marker -- candidate lists, per-candidate (ipass, text) and all message dict fields, through the C ABI."""
    from conftest import check_against_light_golden, load_light_frames
    from pyft8_amd import _lib, messages as M
    from pyft8_amd.receiver import config_from_kwargs
    groups = {}
    for e, audio in load_light_frames():
        groups.setdefault(json.dumps(e["kwargs"], sort_keys=True), []).append((e, audio))
    n_msgs = n_dev = 0
    for key, items in groups.items():
        cfg = config_from_kwargs(**json.loads(key))
        h = _lib.Handle(cfg, max_frames=len(items))
        rec, cnt, ev, evc = h.decode_batch(np.stack([a for _, a in items]))
        h.close()
        for i, (e, audio) in enumerate(items):
            n = int(cnt[i])
            cands = [(int(r["f0_idx"]), int(r["h0_idx"])) for r in rec[i, :n]]
            tab = M.CallHashes()
            outs = []
            for r in rec[i, :n]:
                if int(r["status"]) == 1:
                    t = M.unpack((int(r["msg_hi"]) << 64) | int(r["msg_lo"]), tab)
                    outs.append((int(r["ipass"]), " ".join(t or ())))
                else:
                    outs.append(None)
            msgs = M.package_frame(rec[i], n, ev[i], int(evc[i]), cyclestart_string="700101_000015")
            for (f0, h0, sc), r in zip(e["cands"], rec[i, :n]):
                if "expected_difference" not in e:
                    assert abs(float(r["score"]) - sc) <= 1e-5 * abs(sc)
            n_msgs += check_against_light_golden(e, cands, outs, msgs)
            n_dev += "expected_difference" in e
            if "expected_difference" in e:
                # a marked frame must not hide a deviation of the GPU from the oracle -- on these five frames the
                # records (every candidate's outcome, ipass, AP, method, iteration / trial index, payload) and the messages are
                # the ORACLE's, bit for bit; only the oracle-vs-reference difference is what the marker documents
                assert cfg.f0_hi <= 960                                    # the default-width oracle build applies
                _check_frame(rec[i], cnt[i], ev[i], evc[i], audio, None, O.default_config(**_lib.fft_plans(), **{
                    k: getattr(cfg, k) for k in ("sync_score_min", "max_cands", "f0_lo", "f0_hi", "h0_lo", "h0_hi")}))
    assert n_msgs > 400 and n_dev == 1


def test_wide_build_reference_goldens_and_oracle():
    """search_freq_range beyond 3000 Hz runs on libft8rx_wide.so (same source, wide layouts): the three wide goldens of the REAL
    reference (carriers up to 5.6 kHz) through the C ABI -- candidates, outcomes, every message dict field -- and, stage by stage,
    bit-identical spectrogram / cycle spectrum / whole-frame records against the oracle's wide build; where the two builds
    overlap they agree bit for bit.  Receiver takes the same kwargs and sizes search_grid like the reference (f0_hi + 16 columns)."""
    from conftest import check_against_light_golden, load_wide_frames
    from pyft8_amd import _lib, messages as M
    from pyft8_amd.receiver import Receiver, config_from_kwargs
    n_hi = 0
    for e, audio in load_wide_frames():
        cfg = config_from_kwargs(**e["kwargs"])
        h = _lib.Handle(cfg, max_frames=2)
        assert h.wide and (h.grid_cols, h.spec_bins) == (1920, 96000)
        ocfg = O.default_config(sync_score_min=cfg.sync_score_min, max_cands=cfg.max_cands, f0_lo=cfg.f0_lo, f0_hi=cfg.f0_hi,
                                h0_lo=cfg.h0_lo, h0_hi=cfg.h0_hi)
        assert bits_equal(h.spectrogram(audio)[0], O.spectrogram(audio, ocfg))
        assert bits_equal(h.cycle_spectrum(audio)[0].view(np.float32), O.cycle_spectrum(audio, ocfg).view(np.float32))
        rec, cnt, ev, evc = h.decode_batch(np.stack([audio, audio]))
        h.close()
        assert rec[0].tobytes() == rec[1].tobytes()
        n = _check_frame(rec[0], cnt[0], ev[0], evc[0], audio, None, ocfg)              # records, outcomes, messages == oracle (wide)
        cands = [(int(r["f0_idx"]), int(r["h0_idx"])) for r in rec[0, :cnt[0]]]
        tab = M.CallHashes()
        outs = []
        for r in rec[0, :cnt[0]]:
            outs.append((int(r["ipass"]), " ".join(M.unpack((int(r["msg_hi"]) << 64) | int(r["msg_lo"]), tab) or ())) if int(r["status"]) == 1 else None)
        msgs = M.package_frame(rec[0], int(cnt[0]), ev[0], int(evc[0]), cyclestart_string="700101_000015")
        assert check_against_light_golden(e, cands, outs, msgs) == n == len(e["messages"])
        n_hi += sum(m["fHz"] > 3000 for m in e["messages"])
    assert n_hi >= 30
    # four more frames (dense, up to 5.6 kHz; sparse and weak) in one batch: every record and message equals the oracle's wide build
    from pyft8_amd import synth
    batch = np.stack([synth.make_frame(7200000 + k, n_signals=(60, 60, 6, 6)[k], snr_range=((-6.0, 12.0), (-18.0, 0.0), (-22.0, -10.0), (0.0, 5.0))[k],
                                       freq_range=(150.0, 5650.0)) for k in range(4)])
    cfg = config_from_kwargs(search_freq_range=[100, 5900])
    ocfg = O.default_config(f0_lo=cfg.f0_lo, f0_hi=cfg.f0_hi, h0_lo=cfg.h0_lo, h0_hi=cfg.h0_hi)
    h = _lib.Handle(cfg, max_frames=4)
    rec, cnt, ev, evc = h.decode_batch(batch)
    h.close()
    assert sum(_check_frame(rec[i], cnt[i], ev[i], evc[i], batch[i], None, ocfg) for i in range(4)) > 50
    e, audio = load_wide_frames()[0]
    hd, hw = _lib.Handle(max_frames=1), _lib.Handle(_lib.default_config(f0_hi=1800), max_frames=1)
    assert bits_equal(hd.spectrogram(audio)[0], hw.spectrogram(audio)[0][:, :976])
    assert bits_equal(hd.cycle_spectrum(audio)[0].view(np.float32), hw.cycle_spectrum(audio)[0][:49152].view(np.float32))
    hd.close(); hw.close()
    got = []
    rx = Receiver("x", got.append, **e["kwargs"])
    assert list(rx.audio_in.search_grid.shape) == e["grid_shape"]
    out = rx.decode_frame(audio)
    assert [" ".join(m["msg_tuple"]) for m in out] == [" ".join(m["msg_tuple"]) for m in e["messages"]]
    for m, ref in zip(out, e["messages"]):
        for key in ("tsec", "fHz", "their_snr", "all_txt_format", "tweaks", "decode_notes"):
            assert m[key] == ref[key], (key, m[key], ref[key])
    with pytest.raises(_lib.Ft8rxError):
        Receiver("x", None, search_freq_range=[100, 6000])


def test_decode_batch_synthetic_vs_oracle(H, ocfg):
    from pyft8_amd import synth
    audio = synth.make_batch(1000, 8)
    rec, cnt, ev, evc = H.decode_batch(audio)
    total = 0
    for i in range(len(audio)):
        total += _check_frame(rec[i], cnt[i], ev[i], evc[i], audio[i], None, ocfg)
    assert total > 8 * 15


def test_edge_frames(H, ocfg):
    rng = np.random.default_rng(9)
    silence = np.zeros(180000, np.int16)
    noise = np.clip(np.rint(rng.standard_normal(180000) * 1000), -32768, 32767).astype(np.int16)
    loud = np.clip(np.rint(rng.standard_normal(180000) * 30000), -32768, 32767).astype(np.int16)
    t = np.arange(180000)
    tone = np.rint(12000 * np.sin(2 * np.pi * 1000.0 * t / 12000)).astype(np.int16)            # one bin, exact zeros elsewhere
    nyq = np.where(t % 2 == 0, 32767, -32768).astype(np.int16)                                  # full-scale Nyquist
    dc = np.full(180000, 12345, np.int16)
    impulse = np.zeros(180000, np.int16); impulse[90000] = 32767
    burst = noise.copy(); burst[60000:60480] = 32767                                            # one clipped hop
    half = noise.copy(); half[90000:] = 0                                                       # audio stops mid-frame
    audio = np.stack([silence, noise, loud, tone, nyq, dc, impulse, burst, half])
    rec, cnt, ev, evc = H.decode_batch(audio)
    assert cnt[0] == 0                       # digital silence: grid = -240 dB everywhere, no candidates
    for i in range(1, len(audio)):           # every degenerate input: same candidates, records, events and messages as the oracle
        _check_frame(rec[i], cnt[i], ev[i], evc[i], audio[i], None, ocfg)


def test_receiver_surface_matches_reference_listing():
    """Drop-in surface: Receiver(...).decode_frames + on_message callback, decoders.* functions."""
    from pyft8_amd.receiver import Receiver
    from pyft8_amd import decoders
    got = []
    rx = Receiver("x", got.append)
    audio, gold, js = load_golden("test_09")
    msgs = rx.decode_frame(audio)
    assert [m["all_txt_format"] for m in got] == [m["all_txt_format"] for m in js["messages"]]
    assert msgs == got
    rx.audio_in.load_frame(audio)
    cands = rx.search("700101_000015", 0)
    assert [c.origin["f0_idx"] for c in cands] == list(gold["f0_idx"])
    assert rx.audio_in.waterfall_data["data"].shape == (488, 375)
    k = [i for i, r in enumerate(js["bp_result"]) if r is not None][0]
    llr = gold["bp_llr_in"][k].copy()
    res, nits, out = decoders.ldpc_decode(llr, int(gold["bp_nc0max"][k]), int(gold["bp_iters"][k]))
    assert "<" not in js["bp_result"][k]                     # (a hashed call would need the reference's table state: pick another vector then)
    assert " ".join(res) == js["bp_result"][k]
    k = [i for i, r in enumerate(js["osd_result"]) if r is not None][0]
    assert " ".join(decoders.osd_012(gold["osd_llr_in"][k])) == js["osd_result"][k]
    assert decoders.unpack(int('00000000000000000100011011110000010010000000000111000001100011111000010010001', 2)) == ("CQ DX", "G1OJS", "IO90")


def test_candidate_decode_ladder_like_the_reference_harness():
    """VERDICT r2 missing #5: Candidate.decode() / check_and_package() (reference receiver.py:51-107) exist on the candidates that
    Receiver.search returns, so the reference's OWN driving loop -- SURVEY 8c steps 6-7, i.e. manage_cycle's inner loop
    (receiver.py:389-398): per round every undecoded candidate, llr_sd-descending, advances one ipass -- runs unchanged on this
    package.  On both fixture recordings it emits exactly the reference's message dicts (every key but decode_completed), in order."""
    from pyft8_amd.receiver import Receiver
    from pyft8_amd import decoders as D
    for name in ("test_09", "test_08"):
        audio, gold, js = load_golden(name)
        D.call_hashes.clear()
        got = []
        rx = Receiver("x", got.append)
        rx.audio_in.load_frame(audio)
        cands = rx.search("700101_000015", 0, range(*rx.audio_in.search_f0_idx_range))
        assert len(cands) == js["n_cands"] == len(gold["f0_idx"])
        assert [(c.origin["f0_idx"], c.origin["h0_idx"]) for c in cands] == list(zip(gold["f0_idx"].tolist(), gold["h0_idx"].tolist()))
        dup = set()
        for rnd in range(8):
            for c in sorted([c for c in cands if not c.decode_result], key=lambda c: c.llr_sd, reverse=True):
                c.decode(10 + rnd)
                if c.decode_result not in (None, "stop"):
                    c.check_and_package(dup)
        assert len(got) == len(js["messages"]), (name, len(got), len(js["messages"]))
        for m, ref in zip(got, js["messages"]):
            for key, val in ref.items():
                if key != "decode_completed":
                    assert (list(m[key]) if key == "msg_tuple" else m[key]) == val, (name, key, m[key], val)
        assert all(c.ipass == 8 or c.decode_result == "stop" for c in cands)


def test_search_sub_range_and_second_cycle():
    """Receiver.search(cyclestart_string, odd_even, search_f_idxs) (reference receiver.py:338-367): a contiguous sub-range of
    f0 indices equals the oracle's search with that range; the second half of the 750-row grid (odd_even=1) gives the
    same candidates as the first, with the grid bounds moved by 375 hops."""
    from pyft8_amd import _lib
    from pyft8_amd.receiver import Receiver
    rx = Receiver("x", None)
    audio, gold, js = load_golden("test_08")
    rx.audio_in.load_frame(audio)
    full = rx.search("700101_000015", 0, range(32, 960))
    assert [c.origin["f0_idx"] for c in full] == list(gold["f0_idx"])
    sub = rx.search("700101_000015", 0, range(300, 340))
    ocfg = O.default_config(**_lib.fft_plans(), f0_lo=300, f0_hi=340)
    grid = O.spectrogram(audio, ocfg)
    oc = O.sync_search(grid, ocfg)
    assert len(oc) > 5
    assert [(c.origin["f0_idx"], c.origin["h0_idx"]) for c in sub] == [(int(c.f0_idx), int(c.h0_idx)) for c in oc]
    assert np.array_equal(np.float32([c.origin["score"] for c in sub]), np.float32([c.score for c in oc]))
    assert rx.search("700101_000015", 0, []) == []
    # any index sequence (the reference just iterates over it): runs of consecutive indices, single indices, and a descending
    # tail -- threshold per f0, stable sort by score (ties in iteration order), cut at max_cands
    idx = list(range(40, 200)) + list(range(500, 640)) + [700, 702, 704] + [330, 329, 328]
    got = rx.search("700101_000015", 0, idx)
    ocfg_all = O.default_config(**_lib.fft_plans(), max_cands=2000)
    allc = {int(c.f0_idx): c for c in O.sync_search(O.spectrogram(audio, ocfg_all), ocfg_all)}
    pos = {f: k for k, f in enumerate(idx)}
    want = sorted((allc[f] for f in idx if f in allc), key=lambda c: (-c.score, pos[int(c.f0_idx)]))[:200]
    assert len(want) > 20
    assert [(c.origin["f0_idx"], c.origin["h0_idx"]) for c in got] == [(int(c.f0_idx), int(c.h0_idx)) for c in want]
    assert np.array_equal(np.float32([c.origin["score"] for c in got]), np.float32([c.score for c in want]))
    with pytest.raises(_lib.Ft8rxError):
        rx.search("700101_000015", 0, [300, 300])
    with pytest.raises(_lib.Ft8rxError):
        rx.search("700101_000015", 0, range(950, 970))          # beyond the grid the reference would have (f0_hi + 16 columns)
    # the same hops stored as the grid's second cycle (rows 376..749 and the wrap row 0)
    g = rx.audio_in.search_grid
    rows = g[1:376].copy()
    g[:] = 1.0
    g[376:] = rows[:374]
    g[0] = rows[374]
    odd = rx.search("700101_000030", 1, range(32, 960))
    assert [(c.origin["f0_idx"], c.origin["h0_idx"], c.origin["score"]) for c in odd] == \
           [(c.origin["f0_idx"], c.origin["h0_idx"], c.origin["score"]) for c in full]
    assert [c.search_grid_bounds for c in odd] == [[b[0] + 375, b[1] + 375] for b in (c.search_grid_bounds for c in full)]
    assert odd[0].origin["odd_even"] == 1


def test_ragged_and_empty_batches(ocfg):
    """A frame shorter than 15 s decodes like the same frame padded with silence (oracle on the padded frame); an empty
    batch is an empty result; wrong shapes raise."""
    from pyft8_amd import _lib
    from pyft8_amd.receiver import Receiver, frames_from_ragged
    got = []
    rx = Receiver("x", got.append, max_frames=2)
    audio, gold, js = load_golden("synth_000000")
    short = audio[:150000]                               # 12.5 s: signals starting late lose their tail
    out = rx.decode_frames([short, audio])
    assert [m["all_txt_format"] for m in out[1]] == [m["all_txt_format"] for m in js["messages"]]
    padded = frames_from_ragged([short])[0]
    want = O.decode_frame(padded, ocfg)
    assert [" ".join(m["msg_tuple"]) for m in out[0]] == [" ".join(m["msg_tuple"]) for m in want["msgs"]]
    assert 0 < len(out[0]) <= len(out[1])
    assert rx.decode_frames(np.zeros((0, _lib.NSAMP), np.int16)) == []
    # page-locked host audio (ft8rx_alloc_host): same results as the pageable path
    h = _lib.default_handle(2)
    pin = h.pinned_audio(2)
    pin[:] = np.stack([padded, audio])
    ra, ca, ea, eca = h.decode_batch(pin)
    rb, cb, eb, ecb = h.decode_batch(np.stack([padded, audio]))
    assert np.array_equal(ca, cb) and ra.tobytes() == rb.tobytes() and np.array_equal(eca, ecb)
    del pin
    with pytest.raises(_lib.Ft8rxError):
        rx.decode_frames(np.zeros((1, _lib.NSAMP + 5), np.int16))
    with pytest.raises(_lib.Ft8rxError):
        _lib.default_handle().decode_batch(np.zeros((1, 1000), np.int16))


def test_device_validity_and_crc_on_reference_message_golden(H):
    """Device-side unpack validity (ft8rx_valid77) and CRC-14 (ft8rx_crc_valid) on the reference-generated message golden:
    valid  <=>  the reference's unpack() returned a tuple; CRC of the reference transmitter's codewords passes."""
    from pyft8_amd import synth
    d = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "messages.json")))
    words = [int(u["bits77"], 16) for u in d["unpack"]]
    got = H.valid77(words)
    want = np.array([u["result"] is not None for u in d["unpack"]])
    assert np.array_equal(got != 0, want)
    cws, bits = [], []
    for p in d["pack"]:
        b = int(p["bits77"], 16)
        w91 = (b << 14) | p["crc14"]
        cws.append([1.0 if (w91 >> (90 - i)) & 1 else -1.0 for i in range(91)])
        bits.append(b)
    res, lo, hi = H.crc_valid(np.float32(cws))
    assert np.all(res != 0)                                   # CRC passes on every reference-made codeword
    assert [int(l) | (int(h) << 64) for l, h in zip(lo, hi)] == bits
    bad = np.float32(cws)
    bad[:, 5] *= -1                                           # one flipped message bit: CRC must fail
    assert not H.crc_valid(bad)[0].any()


def test_result_slots_pipeline_order():
    """Double-buffered results (include/ft8rx.h): fetch returns the OLDEST unfetched batch, a third enqueue drops the oldest,
    fetching again without a new enqueue returns the latest batch, and pipelined results equal synchronous ones."""
    from pyft8_amd import _lib
    store = _lib.Handle(max_frames=3)                  # its staging buffer holds three device-resident frames
    base = store.staging_ptr()
    store.synth_frames(base, 91000, 3, n_signals=20, snr_range=(-12.0, 5.0))
    frames = store.download_audio(base, 3)
    h = _lib.Handle(max_frames=1)
    want = [h.decode_batch(frames[i:i + 1]) for i in range(3)]

    class _D:                                          # device pointer of frame i
        def __init__(self, i):
            self.p = base + i * _lib.NSAMP * 2

        def data_ptr(self):
            return self.p
    d = [_D(i) for i in range(3)]

    def same(a, b):                                    # valid parts only: entries past the counts are stale slot contents
        (ra, ca, ea, na), (rb, cb, eb, nb) = a, b
        return (np.array_equal(ca, cb) and np.array_equal(na, nb) and ra[0, :ca[0]].tobytes() == rb[0, :cb[0]].tobytes()
                and sorted(ea[0, :na[0]].tolist()) == sorted(eb[0, :nb[0]].tolist()))
    h.enqueue(d[0].data_ptr(), 1)
    h.enqueue(d[1].data_ptr(), 1)
    assert same(h.fetch(1), want[0])
    h.enqueue(d[2].data_ptr(), 1)                      # slot of batch 0 is reused while batch 1 is still unfetched
    assert same(h.fetch(1), want[1])
    assert same(h.fetch(1), want[2])
    assert same(h.fetch(1), want[2])                   # nothing new: the latest batch again
    for i in range(3):                                 # three enqueues, no fetch: batch 0 is dropped
        h.enqueue(d[i].data_ptr(), 1)
    assert same(h.fetch(1), want[1]) and same(h.fetch(1), want[2])
    with pytest.raises(_lib.Ft8rxError):
        _lib.Handle(max_frames=1).fetch(1)             # nothing enqueued yet
    h.close()
    store.close()


def test_free_running_chunks_with_other_calls_in_flight(ocfg):
    """Round 3: consecutive enqueued batches run as free-running chunk streams (no per-batch fork / join; include/ft8rx.h).  Every
    other entry point first waits for the batches in flight, and a change of batch size or stream count re-synchronises the streams:
    stage calls, the synchronous entry, subtraction and size / stream changes issued BETWEEN an enqueue and its fetch leave every
    batch's results exactly what a quiet handle produces."""
    import hashlib
    from pyft8_amd import _lib
    B = 64
    store = _lib.Handle(max_frames=2 * B)
    base = store.staging_ptr()
    store.synth_frames(base, 93000, 2 * B, n_signals=30, snr_range=(-10.0, 8.0))
    host = store.download_audio(base, 2 * B)

    def digest(res, n):
        rec, cnt, ev, evc = res
        hsh = hashlib.sha256()
        for f in range(n):
            hsh.update(rec[f, :cnt[f]].tobytes())
            hsh.update(np.sort(ev[f, :min(int(evc[f]), _lib.EVENT_CAP)], order=["cand", "ipass", "slot", "seq"]).tobytes())
        return hsh.hexdigest()
    h = _lib.Handle(max_frames=B)
    want = [digest(h.decode_batch(host[k * B:(k + 1) * B]), B) for k in range(2)]
    want_half = digest(h.decode_batch(host[:B // 2]), B // 2)
    ptr = [base, base + B * _lib.NSAMP * 2]
    grid_want = O.spectrogram(host[5], ocfg)
    for rnd in range(3):
        h.enqueue(ptr[0], B)
        h.enqueue(ptr[1], B)                                           # two batches in flight on free-running streams
        g = h.spectrogram(host[5])[0]                                  # a stage call: waits for them, then uses the shared workspaces
        assert bits_equal(g[1:376], grid_want[1:376])
        assert digest(h.fetch(B), B) == want[0] and digest(h.fetch(B), B) == want[1], rnd
        h.enqueue(ptr[0], B)
        h.enqueue(ptr[0], B // 2)                                      # another partition right behind it
        assert digest(h.fetch(B), B) == want[0] and digest(h.fetch(B // 2), B // 2) == want_half, rnd
        h.enqueue(ptr[1], B)
        h.set_streams(4 if rnd % 2 == 0 else 2)                         # takes effect with the next batch
        h.enqueue(ptr[0], B)
        assert digest(h.fetch(B), B) == want[1] and digest(h.fetch(B), B) == want[0], rnd
        h.enqueue(ptr[1], B)
        assert digest(h.decode_batch(host[:B]), B) == want[0]          # the synchronous entry in between drops the batch in flight ...
        h.enqueue(ptr[1], B)
        ok, lo, hi, nits, has, out = h.ldpc(np.zeros((3, 174), np.float32) + 1.0, 35, 5)       # ... and so may any stage call
        assert digest(h.fetch(B), B) == want[1], rnd
    h.close()
    store.close()


def test_pipelined_host_entry_matches_synchronous_decode():
    """ft8rx_enqueue_batch_host (H2D of batch k+1 overlapping the kernels of batch k, two device staging buffers): a stream of
    different batches from page-locked and from pageable host memory gives exactly the results of ft8rx_decode_batch, in order."""
    from pyft8_amd import _lib, synth
    B = 24
    batches = [np.stack([synth.make_frame(52000 + 100 * k + i, n_signals=12, snr_range=(-10.0, 6.0)) for i in range(B)]) for k in range(4)]
    h = _lib.Handle(max_frames=B)
    want = [h.decode_batch(b) for b in batches]

    def same(a, b):
        (ra, ca, ea, na), (rb, cb, eb, nb) = a, b
        if not (np.array_equal(ca, cb) and np.array_equal(na, nb)):
            return False
        return all(ra[f, :ca[f]].tobytes() == rb[f, :cb[f]].tobytes() and
                   sorted(ea[f, :min(na[f], _lib.EVENT_CAP)].tolist()) == sorted(eb[f, :min(nb[f], _lib.EVENT_CAP)].tolist()) for f in range(B))
    for pinned in (True, False):
        bufs = []
        for b in batches:
            a = h.pinned_audio(B) if pinned else np.empty_like(b)
            a[:] = b
            bufs.append(a)
        for rounds in range(2):                                  # steady state: enqueue(k+1) before fetch(k)
            h.enqueue_host(bufs[0])
            for k in range(1, 4):
                h.enqueue_host(bufs[k])
                assert same(h.fetch(B), want[k - 1]), (pinned, rounds, k)
            assert same(h.fetch(B), want[3])
    assert same(h.decode_batch(batches[1]), want[1])             # the synchronous entry still works in between
    with pytest.raises(_lib.Ft8rxError):
        h.enqueue_host(batches[0][:, :100])
    h.close()


def test_subtract_matches_reference_golden_and_oracle():
    """SURVEY 8f-4 primitive: ft8rx_subtract vs the reference's Receiver.subtract_signal run in isolation
    (tests/golden/subtract.npz, oracle/gen_golden_subtract.py) and vs the C oracle's restatement.  Floating-point stage:
    tolerance 1e-4 of the audio's RMS (north star), measured ~1e-6; the int16 residual is the rounded float residual."""
    from pyft8_amd import _lib, synth
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "subtract.npz"))
    pos = g["pos"]
    ncase = len(g["recipes"])
    frames, sigs = [], []
    for ci in range(ncase):
        idx, ns, lo, hi = g["recipes"][ci]
        frames.append(synth.make_frame(int(idx), n_signals=int(ns), snr_range=(lo, hi)))
        sigs.append([(t, float(f), float(ts)) for t, f, ts in zip(g[f"c{ci}_tones"], g[f"c{ci}_fHz"], g[f"c{ci}_tsec"])])
    frames = np.stack(frames)
    h = _lib.Handle(max_frames=ncase)
    h.decode_batch(frames)                                   # leaves the frames in the handle's device staging buffer
    ptr = h.staging_ptr()
    res = h.subtract(ptr, ncase, sigs, return_float=True)
    back = h.download_audio(ptr, ncase)
    for ci in range(ncase):
        rms = float(g[f"c{ci}_stats"][0])
        assert np.abs(res[ci][pos] - g[f"c{ci}_after"]).max() < 1e-4 * rms, ci          # vs the reference
        want = frames[ci].astype(np.float32)
        for t, f, ts in sigs[ci]:
            O.subtract(want, t, f, ts)
        assert np.abs(res[ci] - want).max() < 1e-4 * rms, ci                              # vs the oracle, every sample
        assert np.array_equal(back[ci], np.clip(np.rint(res[ci]), -32768, 32767).astype(np.int16))
    assert np.array_equal(back[3], frames[3])                # case 3: int(12000 tsec) = 0 -> nothing subtracted
    assert abs(float(res[0].astype(np.float64).std()) - float(g["c0_stats"][1])) < 0.01
    h.close()


def test_multi_pass_decode_with_subtraction():
    """Extension: decode -> subtract every decoded signal (origin refined on the GPU) -> decode the residual.
    On device-generated frames with known truth: the refined origins hit the true start within 3 ms / 0.1 Hz, pass 1 is
    unchanged, and the second pass adds several true decodes per frame without flooding false ones."""
    from pyft8_amd import _lib, synth
    from pyft8_amd.receiver import Receiver
    n = 12
    got = []
    rx = Receiver("", got.append, max_frames=n)
    h = rx._handle(n)
    truth = h.synth_frames(h.staging_ptr(), 8200000, n, n_signals=50, snr_range=(-10.0, 10.0))
    audio = h.download_audio(h.staging_ptr(), n)
    want = [{t["msg"]: t for t in truth[f]} for f in range(n)]
    one = rx.decode_frames(audio)
    got.clear()
    two = rx.decode_frames(audio, passes=2)
    assert sum(len(x) for x in two) == len(got)                                    # every message went through on_message
    t1 = t2 = f2 = 0
    for f in range(n):
        assert [d["all_txt_format"] for d in two[f][:len(one[f])]] == [d["all_txt_format"] for d in one[f]]     # pass 1 untouched
        extra = two[f][len(one[f]):]
        assert all(d["decode_notes"].endswith("_SUB") for d in extra) and not any(d["decode_notes"].endswith("_SUB") for d in one[f])
        t1 += sum(" ".join(d["msg_tuple"]) in want[f] for d in one[f])
        t2 += sum(" ".join(d["msg_tuple"]) in want[f] for d in two[f])
        f2 += sum(" ".join(d["msg_tuple"]) not in want[f] for d in two[f])
    assert t2 >= t1 + 5 * n and f2 <= 2 * n, (t1, t2, f2)
    # origin refinement against truth
    h.decode_batch(audio)
    sigs, tr = [], []
    for f in range(n):
        keep = [d for d in one[f] if " ".join(d["msg_tuple"]) in want[f] and int(d["their_snr"]) > -10]
        tr.append([want[f][" ".join(d["msg_tuple"])] for d in keep])
        sigs.append([(synth.tones79(synth.pack77(*d["msg_tuple"])), d["fHz"], d["tsec"]) for d in keep])
    resid = {}
    for mode in (1, 2):                     # 1 = full-rate scans, 2 = the decimated-baseband path Receiver uses (kernels/subtract.hpp)
        h.decode_batch(audio)
        res, orig = h.subtract(h.staging_ptr(), n, sigs, refine=mode, return_origins=True, return_float=True)
        dt = np.array([o[1] - t["t0"] for f in range(n) for o, t in zip(orig[f], tr[f])])
        df = np.array([o[0] - t["f0"] for f in range(n) for o, t in zip(orig[f], tr[f])])
        assert len(dt) > 15 * n and np.abs(dt).max() < 0.003 and np.abs(df).max() < 0.1, (mode, np.abs(dt).max(), np.abs(df).max())
        resid[mode] = float(res.astype(np.float64).std())
    rms_in = float(audio.astype(np.float64).std())
    assert resid[1] < 0.7 * rms_in and abs(resid[2] - resid[1]) < 0.01 * resid[1], (resid, rms_in)      # both cancel equally well
    assert rx.subtract_refine == 2
    # edge cases: nothing to subtract, a single frame, more passes than there is anything to find, a real recording, late signals
    rx1 = Receiver("", None)
    rng = np.random.default_rng(4)
    noise = np.clip(np.rint(rng.standard_normal(180000) * 1000), -32768, 32767).astype(np.int16)
    assert rx1.decode_frames(noise, passes=3) == [[]]
    wav, gold, js = load_golden("test_09")
    p1, p5 = rx1.decode_frames(wav), rx1.decode_frames(wav, passes=5)
    assert [d["all_txt_format"] for d in p5[0][:len(p1[0])]] == [d["all_txt_format"] for d in p1[0]] and len(p5[0]) > len(p1[0])
    m, c, rec, cnt = rx1.decode_frames_arrays(wav, passes=2)
    assert c[0] > len(p1[0]) and set(m[0, :c[0]]["pad"][:, 0].tolist()) == {0, 1}
    late = synth.tones_to_wave(synth.tones79(synth.pack77("CQ", "K1ABC", "FN42")), 1500.0)
    x = noise.astype(np.float64)
    x[30000:] += 3000.0 * late[:150000]                                        # starts at 2.5 s and is cut off by the end of the frame
    lf = np.clip(np.rint(x), -32768, 32767).astype(np.int16)
    out = rx1.decode_frames(lf, passes=2)                                      # decodes; the subtraction guard skips it; no crash
    assert [" ".join(d["msg_tuple"]) for d in out[0]][:1] == ["CQ K1ABC FN42"]


def test_refine3_matches_reference_golden_and_oracle(ocfg):
    """ft8rx_subtract(refine = 3) = the experiment's Candidate.refine_time_origin (receiver_sub.py:58-72) + subtract_signal
    (:380-402) per signal on the running residual, against the REAL reference's outputs (tests/golden/refine_time_origin.npz) and the
    oracle: re-estimated origins identical (bit-exact scores: same FFT contract as k_fine), residual within 1e-4 of the audio RMS."""
    from pyft8_amd import _lib, synth
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "refine_time_origin.npz"))
    ncase = len(g["recipes"])
    frames, sigs = [], []
    for ci in range(ncase):
        idx, ns, lo, hi, n = g["recipes"][ci]
        frames.append(synth.make_frame(int(idx), n_signals=int(ns), snr_range=(lo, hi)))
        sigs.append([(t, float(f), float(ts)) for t, f, ts in zip(g[f"c{ci}_tones"], g[f"c{ci}_fHz_in"], g[f"c{ci}_tsec_in"])])
    frames = np.stack(frames)
    h = _lib.Handle(max_frames=ncase)
    h.decode_batch(frames)
    res, orig = h.subtract(h.staging_ptr(), ncase, sigs, refine=3, return_origins=True, return_float=True)
    for ci in range(ncase):
        want = [(float(f), float(t)) for f, t in zip(g[f"c{ci}_fHz_out"], g[f"c{ci}_tsec_out"])]
        assert orig[ci] == want, (ci, orig[ci], want)                                                   # = the reference
        rms = float(g[f"c{ci}_stats"][0])
        assert np.abs(res[ci][g["pos"]] - g[f"c{ci}_after"]).max() < 1e-4 * rms, ci                      # vs the reference's buffer
        wf = frames[ci].astype(np.float32)
        for t, f, ts in sigs[ci]:
            f2, t2, _ = O.refine_time_origin(wf, f, ts, ocfg)
            O.subtract(wf, t, f2, t2)
        assert np.abs(res[ci] - wf).max() < 1e-4 * rms, ci                                              # vs the oracle, every sample
    h.close()


def test_multi_pass_matches_the_oracle_composition(ocfg):
    """VERDICT r2 #2 (f-4 above the primitive): Receiver.decode_frames_arrays(passes=2) against the oracle's restatement of the same
    composition (oracle.decode_frame_passes: decode -> subtraction list -> refine = 2 re-estimation on the decimated baseband copy ->
    subtraction -> int16 -> decode -> merge) on 8 config-1 frames and the reference's test_09.wav.  The re-estimated origins are
    arg-maxima on a (2.67 ms, 1/64 Hz) grid computed in float32 with hardware sin/cos on the GPU and in double in the oracle:
    stated tolerance = the same grid point for >= 97 % of the signals and never more than one step off; the message lists
    (pass index, text) are identical in every frame whose origins all agree, and in at least 8 of the 9 frames."""
    from pyft8_amd import _lib, synth
    from pyft8_amd.receiver import Receiver
    n = 8
    rx = Receiver("", None, max_frames=n + 1)
    h = rx._handle(n + 1)
    h.synth_frames(h.staging_ptr(), 8300000, n, n_signals=50, snr_range=(-10.0, 10.0))
    audio = np.concatenate([h.download_audio(h.staging_ptr(), n), load_golden("test_09")[0][None]])
    B = n + 1
    msgs, mcnt, rec, cnt = rx.decode_frames_arrays(audio, passes=2)
    # the GPU's refined origins of the first sweep: the same call sequence by hand
    r1, c1, e1, ec1 = h.decode_batch(audio)
    m1, mc1 = _lib.package_batch(r1, c1, e1, ec1)
    sl = _lib.subtraction_list(m1, mc1, r1, -10)
    _, orig = h.subtract(h.staging_ptr(), B, sl, refine=2, return_origins=True)
    same_pt = tot = 0
    frames_equal = 0
    for f in range(B):
        want = O.decode_frame_passes(audio[f], ocfg, passes=2)
        mine = [(int(m["pad"][0]), tuple(x.decode() for x in m["f"])) for m in msgs[f, :mcnt[f]]]
        theirs = [(p, m["msg_tuple"]) for p, m in want["msgs"]]
        assert [t for p, t in mine if p == 0] == [t for p, t in theirs if p == 0], f
        assert len(orig[f]) == len(want["origins"][0]), f
        all_same = True
        for (fg, tg), (fo, to) in zip(orig[f], want["origins"][0]):
            ds, dfq = round((tg - to) * 12000), (fg - fo) * 64
            assert abs(ds) <= 32 and abs(dfq) <= 1.0 + 1e-6, (f, tg - to, fg - fo)
            ok = ds == 0 and abs(dfq) < 1e-6
            same_pt += ok; tot += 1; all_same &= ok
        if all_same:
            assert mine == theirs, (f, mine, theirs)
        frames_equal += mine == theirs
    assert tot > 150 and same_pt >= 0.97 * tot, (same_pt, tot)
    assert frames_equal >= B - 1, frames_equal
    print(f"multi-pass vs oracle: {same_pt}/{tot} refined origins on the same grid point, {frames_equal}/{B} frames with identical message lists")


def test_local_research_mask_and_composition(ocfg):
    """The experiment's re-search (tests/pipeline/receiver_sub.py:434-445: after a subtraction, search(f0 - 2 .. f0 + 1,
    ignore_sync_score_min = True)) as ft8rx_set_search_mask: (1) a batch decoded under a mask -- only the masked columns, every
    score above 0 -- equals the oracle under the same mask record for record, and the mask is gone after set_search_mask(None);
    (2) Receiver.decode_frames_arrays(passes=2, research="local") against oracle.decode_frame_passes(research="local"): identical
    (pass, text) lists in every frame whose refined origins all agree, and in at least 5 of 6 frames."""
    from pyft8_amd import _lib, synth
    from pyft8_amd.receiver import Receiver
    n = 5
    rx = Receiver("", None, max_frames=n + 1)
    h = rx._handle(n + 1)
    h.synth_frames(h.staging_ptr(), 8400000, n, n_signals=50, snr_range=(-10.0, 10.0))
    audio = np.concatenate([h.download_audio(h.staging_ptr(), n), load_golden("test_08")[0][None]])
    B = n + 1
    # (1) masked search, bit for bit
    rng = np.random.default_rng(3)
    mask = (rng.random((B, rx.cfg.f0_hi - rx.cfg.f0_lo)) < 0.08).astype(np.uint8)
    mask[1] = 0                                                     # a frame with nothing to search
    h.set_search_mask(mask)
    rec, cnt, ev, evc = h.decode_batch(audio)
    h.set_search_mask(None)
    assert cnt[1] == 0 and cnt.max() <= int(mask.sum(axis=1).max())
    for f in range(B):
        O.set_search_mask(mask[f])
        try:
            _check_frame(rec[f], cnt[f], ev[f], evc[f], audio[f], None, ocfg)
        finally:
            O.set_search_mask(None)
        assert all(mask[f, int(r["f0_idx"]) - rx.cfg.f0_lo] for r in rec[f, :cnt[f]])
    rec0, cnt0, ev0, evc0 = h.decode_batch(audio)
    for f in range(2):
        _check_frame(rec0[f], cnt0[f], ev0[f], evc0[f], audio[f], None, ocfg)               # the configured search is back
    # (2) the composition
    msgs, mcnt, _, _ = rx.decode_frames_arrays(audio, passes=2, research="local")
    full, fcnt, _, _ = rx.decode_frames_arrays(audio, passes=2)
    r1, c1, e1, ec1 = h.decode_batch(audio)
    m1, mc1 = _lib.package_batch(r1, c1, e1, ec1)
    _, orig = h.subtract(h.staging_ptr(), B, _lib.subtraction_list(m1, mc1, r1, -10), refine=2, return_origins=True)
    frames_equal = n_local = n_full = 0
    for f in range(B):
        want = O.decode_frame_passes(audio[f], ocfg, passes=2, research="local")
        mine = [(int(m["pad"][0]), tuple(x.decode() for x in m["f"])) for m in msgs[f, :mcnt[f]]]
        theirs = [(p, m["msg_tuple"]) for p, m in want["msgs"]]
        assert [t for p, t in mine if p == 0] == [t for p, t in theirs if p == 0], f
        all_same = all(round((tg - to) * 12000) == 0 and abs((fg - fo) * 64) < 1e-6 for (fg, tg), (fo, to) in zip(orig[f], want["origins"][0]))
        if all_same:
            assert mine == theirs, (f, mine, theirs)
        frames_equal += mine == theirs
        n_local += sum(p == 1 for p, _ in mine)
        n_full += int((full[f, :fcnt[f]]["pad"][:, 0] == 1).sum())
    assert frames_equal >= B - 1, frames_equal
    # (±2 search columns = ±6 Hz around a 50-Hz-wide subtracted signal: most of what a subtraction uncovers lies further away, so the
    # experiment's local re-search finds only part of what a full second pass finds -- 15 vs 54 messages on these six frames)
    assert 0 < n_local <= n_full, (n_local, n_full)
    with pytest.raises(_lib.Ft8rxError, match="research"):
        rx.decode_frames(audio[:1], passes=2, research="nearby")
    print(f"local re-search vs oracle: {frames_equal}/{B} frames with identical message lists; second-pass messages: local {n_local}, full {n_full}")


def test_special_message_types_through_the_pipeline(H, ocfg):
    """Frames carrying i3 = 4 / hashed / suffixed / directed-CQ messages (words from the reference-generated message golden):
    every record, event and rendered message equals the oracle's, and the special forms do come out."""
    from conftest import special_message_words
    from pyft8_amd import synth, messages as M
    words = special_message_words()
    assert len(words) >= 20
    audio = np.stack([synth.frame_from_words(0, words), synth.frame_from_words(1, words[::-1], snr_range=(-8.0, 2.0))])
    rec, cnt, ev, evc = H.decode_batch(audio)
    texts = []
    for i in range(2):
        _check_frame(rec[i], cnt[i], ev[i], evc[i], audio[i], None, ocfg)
        texts += [" ".join(m["msg_tuple"]) for m in M.package_frame(rec[i], int(cnt[i]), ev[i], int(evc[i]))]
    assert any("<" in t for t in texts) and any("/P" in t for t in texts) and any("/R" in t for t in texts)
    assert sum(t.startswith("CQ ") for t in texts) >= 3


def _decode_with(cfg_kw, audio):
    from pyft8_amd import _lib
    cfg = _lib.default_config(**cfg_kw)
    h = _lib.Handle(cfg, max_frames=len(audio))
    try:
        return h.decode_batch(audio)
    finally:
        h.close()


def test_large_batch_decode_set_equality():
    """96 fresh synthetic frames (config-1 recipe): every candidate record and every message identical to the oracle."""
    from pyft8_amd import _lib, synth
    audio = synth.make_batch(50000, 96)
    rec, cnt, ev, evc = _decode_with({}, audio)
    ocfg = O.default_config(**_lib.fft_plans())
    total = 0
    for i in range(len(audio)):
        total += _check_frame(rec[i], cnt[i], ev[i], evc[i], audio[i], None, ocfg)
    assert total > 96 * 20


def test_extension_knobs_pipeline():
    """BASELINE config 2/4 knobs (BP 30 iterations, OSD 40 single / 3 double flips): no reference counterpart,
    the oracle run with the same knobs is the checker."""
    from pyft8_amd import _lib, synth
    audio = np.stack([synth.make_frame(70000 + i, n_signals=30, snr_range=(-20.0, -8.0)) for i in range(12)])
    for kw in (dict(bp_iters_b=30, osd_single=40, osd_double=3),
               dict(osd_single=91, osd_double=4),                         # every basis position as a flip row (two-word flip storage)
               dict(bp_iters_b=30, osd_triple=20),                        # order-3 reprocessing, reference acceptance rule
               dict(osd_triple=30, osd_max_hd=32)):                       # config 4 as run by bench.py --config 4: order 3 + distance gate
        rec, cnt, ev, evc = _decode_with(kw, audio)
        ocfg = O.default_config(**_lib.fft_plans(), **kw)
        for i in range(len(audio)):
            _check_frame(rec[i], cnt[i], ev[i], evc[i], audio[i], None, ocfg)


def test_low_snr_frames():
    """Config-4 style stress: few weak signals (-24..-18 dB); OSD does most of the decoding."""
    from pyft8_amd import _lib, synth
    audio = np.stack([synth.make_frame(90000 + i, n_signals=10, snr_range=(-24.0, -18.0)) for i in range(12)])
    rec, cnt, ev, evc = _decode_with({}, audio)
    ocfg = O.default_config(**_lib.fft_plans())
    for i in range(len(audio)):
        _check_frame(rec[i], cnt[i], ev[i], evc[i], audio[i], None, ocfg)


def test_reduced_search_ranges():
    """Non-default Receiver kwargs (pyft8.py:136 intent: sync_score_min=100, max_cands=150) and a narrower search window."""
    from pyft8_amd import _lib, synth
    from pyft8_amd.receiver import config_from_kwargs
    audio = synth.make_batch(60000, 4)
    cfg = config_from_kwargs(sync_score_min=100, max_cands=150, search_freq_range=(300, 2500), search_time_range=(-1.0, 2.0))
    h = _lib.Handle(cfg, max_frames=4)
    rec, cnt, ev, evc = h.decode_batch(audio)
    h.close()
    ocfg = O.default_config(**_lib.fft_plans(), sync_score_min=100.0, max_cands=150, f0_lo=cfg.f0_lo, f0_hi=cfg.f0_hi,
                            h0_lo=cfg.h0_lo, h0_hi=cfg.h0_hi)
    for i in range(4):
        _check_frame(rec[i], cnt[i], ev[i], evc[i], audio[i], None, ocfg)


def test_wide_time_window_clamped_symbols():
    """search_time_range far beyond the reference's default: candidates whose first symbols start before sample 0 (up to all of the
    first Costas block and more) or whose last symbols start beyond sample 3168 of the fine-sync series -- the reference clamps those
    reads (receiver.py:189-195), the frequency-domain grid takes their rows from the clamp position.  Two windows, because the span is
    limited to 352 hops; records, events and LLR-dependent outcomes identical to the oracle."""
    from pyft8_amd import _lib, synth
    from pyft8_amd.receiver import config_from_kwargs
    audio = synth.make_batch(61000, 3)
    for tr in ((-6.0, 3.0), (-1.0, 8.2)):
        cfg = config_from_kwargs(sync_score_min=70, max_cands=256, search_time_range=tr)
        assert _lib.MIN_H0 <= cfg.h0_lo < cfg.h0_hi <= _lib.MAX_H0
        h = _lib.Handle(cfg, max_frames=3)
        rec, cnt, ev, evc = h.decode_batch(audio)
        h.close()
        ocfg = O.default_config(**_lib.fft_plans(), sync_score_min=70.0, max_cands=256, f0_lo=cfg.f0_lo, f0_hi=cfg.f0_hi, h0_lo=cfg.h0_lo, h0_hi=cfg.h0_hi)
        n_edge = n_boundary = 0
        for i in range(3):
            _check_frame(rec[i], cnt[i], ev[i], evc[i], audio[i], None, ocfg)
            r = rec[i][:cnt[i]]
            h0 = r["h0_idx"].astype(np.int64)
            n_edge += int(((h0 < -4) | (h0 > 84)).sum())
            # a symbol starting EXACTLY at the clamp position 3168 with later symbols behind it (round 6: it takes the clamp row, bit-identical
            # to theirs as in the reference): first sample of symbol 0 = 8 h0 (+ 1 for h0 < 0) + ttweak, a multiple of 32 beyond 672
            tb = 8 * h0 + (h0 < 0) + r["ttweak"].astype(np.int64)
            n_boundary += int(((r["nsync"] > 0) & (tb % 32 == 0) & (tb > 672) & (tb <= 3168)).sum())      # nsync > 0: the fine sync ran
        assert n_edge > 50, n_edge           # the window really produced candidates with clamped symbols
        if tr[1] > 8.0:
            assert n_boundary >= 3, n_boundary       # ... and some with the boundary symbol among them


def test_search_time_range_beyond_the_fine_sync_series():
    """The reference takes any search_time_range its 750-row grid can be indexed with (receiver.py:312, 319, 346-347: -36.4 .. +22.6 s) and
    clamps every symbol read (:189-195).  Candidates whose middle Costas block leaves the 3200-sample series (h0 outside [-140, 220]) are
    scored in the time domain with clamped reads (k_fine_td; oracle: ft8o_fine's far_out branch, pinned to the live reference by the
    'far' set of tests/test_reference_crosscheck.py); windows wider than 352 hops run the sync search in several launches.  Four windows
    up to the reference's own limits: every record, event and message equals the oracle's, and the batches really hold far-out
    candidates on both sides, some of which pass the Costas gate."""
    from pyft8_amd import _lib, synth
    from pyft8_amd.receiver import config_from_kwargs
    audio = synth.make_batch(61500, 3)
    n_far = n_far_alive = 0
    for tr in ((-20.0, 20.0), (-30.0, 3.0), (0.0, 22.5), (-36.4, 22.6)):
        cfg = config_from_kwargs(sync_score_min=70, search_time_range=tr)
        assert _lib.MIN_H0 <= cfg.h0_lo < cfg.h0_hi <= _lib.MAX_H0
        h = _lib.Handle(cfg, max_frames=3)
        rec, cnt, ev, evc = h.decode_batch(audio)
        h.close()
        ocfg = O.default_config(**_lib.fft_plans(), sync_score_min=70.0, f0_lo=cfg.f0_lo, f0_hi=cfg.f0_hi, h0_lo=cfg.h0_lo, h0_hi=cfg.h0_hi)
        for i in range(3):
            _check_frame(rec[i], cnt[i], ev[i], evc[i], audio[i], None, ocfg)
            r = rec[i][:cnt[i]]
            far = (r["h0_idx"] < _lib.MIN_H0_FD) | (r["h0_idx"] > _lib.MAX_H0_FD)
            n_far += int(far.sum())
            n_far_alive += int((far & (r["nsync"] > 6)).sum())
    assert n_far > 100 and n_far_alive >= 1, (n_far, n_far_alive)


def test_candidate_cap_and_kwarg_limits():
    """The build's boundary limits against the reference's open-ended kwargs (receiver.py:311-313, 319, 366-367):
    max_cands = 256, the most the default layouts hold -- with sync_score_min = 40 the threshold admits far more than 256 maxima, the
    stable top-K cut really happens, and every record and message still equals the oracle's; beyond the limits that remain Receiver
    names the kwarg instead of failing with the library's generic bad-config error.  (max_cands itself has no limit any more:
    test_max_cands_beyond_256_uses_the_deep_layouts.)"""
    from pyft8_amd import _lib, synth
    from pyft8_amd.receiver import Receiver, config_from_kwargs
    audio = synth.make_batch(64000, 3, n_signals=60, snr_range=(-14.0, 6.0))
    cfg = config_from_kwargs(sync_score_min=40, max_cands=256)
    h = _lib.Handle(cfg, max_frames=3)
    rec, cnt, ev, evc = h.decode_batch(audio)
    h.close()
    assert rec.shape[1] == 256 and (cnt == 256).all(), cnt                # the cap is hit in every frame
    ocfg = O.default_config(**_lib.fft_plans(), sync_score_min=40.0, max_cands=256)
    for i in range(3):
        _check_frame(rec[i], cnt[i], ev[i], evc[i], audio[i], None, ocfg)
    assert float(rec[0]["score"][255]) > 40.0 + 1.0                        # the list ends at the cut, not at the threshold
    for kw, word in ((dict(max_cands=0), "max_cands=0"),
                     (dict(search_time_range=[-37.0, 3.0]), "search_time_range"), (dict(search_time_range=[-6.0, 22.7]), "search_time_range"),
                     (dict(search_time_range=[3.0, 3.0]), "search_time_range"),
                     (dict(search_freq_range=[0, 3000]), "search_freq_range"), (dict(search_freq_range=[100, 6000]), "search_freq_range")):
        with pytest.raises(_lib.Ft8rxError, match=word):
            Receiver("x", None, **kw)


def test_max_cands_beyond_256_uses_the_deep_layouts():
    """max_cands is open-ended in the reference (receiver.py:311-313, 366-367): every f0 bin of the search range whose best sync score
    passes sync_score_min is a candidate -- up to 928 at the default range.  More than 256 select libft8rx_wide.so (FT8RX_MAX_CANDS =
    2048 there, include/ft8rx.h).  Receiver(max_cands=600, sync_score_min=30): more than 256 candidates in EVERY frame, each record,
    event and message equal to the oracle's; the packed form (k_pack_*: 32 mask words per frame in this build) renders the same
    messages; a max_cands beyond the number of f0 bins keeps exactly the list of max_cands = number of bins."""
    from pyft8_amd import _lib, synth
    from pyft8_amd.receiver import Receiver, config_from_kwargs, decode_frames
    audio = synth.make_batch(64100, 3, n_signals=60, snr_range=(-14.0, 6.0))
    cfg = config_from_kwargs(sync_score_min=30, max_cands=600)
    assert cfg.max_cands == 600
    h = _lib.Handle(cfg, max_frames=3)
    assert h.wide
    rec, cnt, ev, evc = h.decode_batch(audio)
    assert rec.shape[1] == 600 and (cnt > 256).all(), cnt
    ocfg = O.default_config(**_lib.fft_plans(), sync_score_min=30.0, max_cands=600)
    for i in range(3):
        _check_frame(rec[i], cnt[i], ev[i], evc[i], audio[i], None, ocfg)
    # packed results of the same batch (page-locked host buffers): byte for byte the numpy twin's, and the same messages as the dense arrays
    cap = _lib.packed_capacity(3, cfg.max_cands)
    pin = [h.pinned_bytes(cap) for _ in range(2)]
    h.set_packed_output(pin[0].ctypes.data, pin[1].ctypes.data, cap)
    h.enqueue_host(audio)
    res = h.fetch(3)
    which, hdr = h.packed_results()
    assert not hdr["overflow"] and hdr["max_cands"] == 600
    assert pin[which][:hdr["bytes"]].tobytes() == _lib.pack_results(*res).tobytes()
    m1, m2 = _lib.package_batch(*res), _lib.package_packed(pin[which][:hdr["bytes"]])
    assert m1[0].tobytes() == m2[0].tobytes() and np.array_equal(m1[1], m2[1]) and (m1[1] > 0).all()
    h.close()
    # no limit: beyond the 928 f0 bins of the default range nothing changes (and the count is whatever passes the threshold)
    big = config_from_kwargs(sync_score_min=30, max_cands=10**9)
    assert big.max_cands == 928
    h = _lib.Handle(big, max_frames=1)
    rec2, cnt2, ev2, evc2 = h.decode_batch(audio[:1])
    h.close()
    assert 600 < int(cnt2[0]) <= 928
    _check_frame(rec2[0], cnt2[0], ev2[0], evc2[0], audio[0], None, O.default_config(**_lib.fft_plans(), sync_score_min=30.0, max_cands=928))
    assert np.array_equal(rec2[0, :600][["f0_idx", "h0_idx", "score"]], rec[0, :600][["f0_idx", "h0_idx", "score"]])
    msgs = decode_frames(audio[:1], sync_score_min=30, max_cands=600)
    assert len(msgs) == 1 and len(msgs[0]) > 0
    rx = Receiver("x", None, max_cands=600, sync_score_min=30)
    assert rx.cfg.max_cands == 600


def test_device_synth_generator(H, ocfg):
    """frames generated on the GPU are valid FT8 (most truth messages decode, none false), have the
    right noise level, are deterministic, and decode identically on GPU and oracle."""
    from pyft8_amd import messages as M
    n = 6
    ptr = H.staging_ptr()
    truth = H.synth_frames(ptr, 123456, n, n_signals=50, snr_range=(-10.0, 10.0))
    audio = H.download_audio(ptr, n)
    truth2 = H.synth_frames(ptr, 123456, n, n_signals=50, snr_range=(-10.0, 10.0))
    assert truth == truth2 and np.array_equal(audio, H.download_audio(ptr, n))
    H.synth_frames(ptr, 999, 1, n_signals=0)
    noise = H.download_audio(ptr, 1)[0].astype(np.float64)
    assert abs(noise.std() - 1000.0) < 10.0 and abs(noise.mean()) < 10.0
    assert abs(np.mean(noise[1:] * noise[:-1])) < 1000.0 ** 2 * 0.02          # white
    rec, cnt, ev, evc = H.decode_batch(audio)
    hits = 0
    for i in range(n):
        _check_frame(rec[i], cnt[i], ev[i], evc[i], audio[i], None, ocfg)
        got = {" ".join(m["msg_tuple"]) for m in M.package_frame(rec[i], int(cnt[i]), ev[i], int(evc[i]))}
        want = {t["msg"] for t in truth[i]}
        assert len(got - want) <= 3                    # the reference algorithm's own OSD false positives (about 1/frame)
        hits += len(got & want)
    assert hits >= n * 20                              # the numpy generator gives ~30 of 50 at this density


def test_device_generator_matches_numpy_twin_sample_for_sample(H):
    """SURVEY 8f-1 / VERDICT r1 #7: k_synth against its numpy twins (pyft8_amd/synth.py) on the same frame indices -- the Philox4x32-10
    + Box-Muller noise alone, the GFSK signal part alone (ft8rx_synth_frames_ex no_noise), and whole frames: equal to +-1 count
    (libm vs device sin/cos/log in the last ulp), differing in < 0.5 % of the samples.  The twin's waveform model is pinned to the
    reference transmitter by tests/test_synth.py::test_waveform_model_matches_reference_transmitter_golden."""
    from pyft8_amd import _lib, synth
    ptr = H.staging_ptr()
    start, n, nsig = 31000, 3, 7
    for noise, sigs in ((True, 0), (False, nsig), (True, nsig)):
        truth, table = H.synth_frames(ptr, start, n, n_signals=sigs, snr_range=(-5.0, 12.0), noise=noise, return_table=True)
        got = H.download_audio(ptr, n).astype(np.int32)
        for f in range(n):
            want = synth.device_frame(start + f, table[f], sigs, noise=noise).astype(np.int32)
            diff = np.abs(got[f] - want)
            assert diff.max() <= 1, (noise, sigs, f, int(diff.max()))
            assert (diff != 0).mean() < 5e-3, (noise, sigs, f, float((diff != 0).mean()))
        if sigs and not noise:
            assert np.abs(got).max() > 500                               # there is a signal


def test_streaming_audio_in_mode(H, ocfg):
    """hop-by-hop AudioIn._callback (PortAudio signature) under a virtual clock: the live search grid
    equals the batch spectrogram row for row, the waterfall view updates in place, and each completed cycle decodes
    to the same messages as the frame-complete path -- for two consecutive cycles (both grid halves)."""
    from pyft8_amd.receiver import Receiver
    vt = [0.0]
    got = []
    rx = Receiver("x", got.append, time_source=lambda: vt[0], early_decode_hop=None)      # one decode per cycle: the golden's emit order
    wf = rx.audio_in.waterfall_data["data"]
    for cyc, name in enumerate(["test_09", "test_08"]):
        audio, gold, js = load_golden(name)
        base = 15.0 * cyc
        for k in range(375):
            vt[0] = base + (k + 1) * 0.04
            rx.audio_in._callback(audio[480 * k:480 * k + 480].tobytes(), 480, None, None)
            assert rx.poll() == [] or k == 374
        want_grid = O.spectrogram(audio, ocfg)
        rows = rx.audio_in.search_grid[375 * cyc + 1:375 * cyc + 376] if cyc == 0 else \
            np.concatenate([rx.audio_in.search_grid[376:750], rx.audio_in.search_grid[0:1]])
        if cyc == 0:
            assert bits_equal(rows, want_grid[1:376])
        else:
            # second cycle: its first 7 hops still see the previous cycle's samples in the 3840-sample window (live ring)
            assert bits_equal(rows[7:], want_grid[8:376])
        assert wf.base is rx.audio_in.search_grid or np.shares_memory(wf, rx.audio_in.search_grid)
        msgs = [m for m in got if True]
        txt = [" ".join(m["msg_tuple"]) for m in got]
        ref_txt = [" ".join(m["msg_tuple"]) for m in js["messages"]]
        if cyc == 0:
            assert txt == ref_txt
            assert got[0]["cyclestart_string"] == "700101_000000" and got[0]["their_tx_cycle"] == 0
        else:
            # same frame samples => same decode set as the frame-complete reference run, except that the stream keeps ONE call-hash
            # table across cycles like the reference's process-global databases.call_hashes (databases.py:8): OR18OSB was heard in
            # cycle 1, so its hash resolves in cycle 2 where the isolated-frame golden prints <...>
            mine = txt[len(txt) - len(ref_txt):]
            assert len(mine) == len(ref_txt)
            resolved = 0
            for a, b in zip(mine, ref_txt):
                if a != b:
                    assert "<...>" in b and a.split(" ")[1:] == b.split(" ")[1:] and a.startswith("<") and "..." not in a, (a, b)
                    resolved += 1
            assert "<OR18OSB> DL8RCH JN68" in mine and resolved >= 1
            assert got[-1]["cyclestart_string"] == "700101_000015" and got[-1]["their_tx_cycle"] == 1
    assert rx.audio_in.cycles_completed == 2


def test_streaming_receiver_runs_itself_like_the_stock_cli():
    """SURVEY 8f-3 / VERDICT r1 #6: the call pattern of the reference CLI (pyft8.py:136-157: construct the Receiver with an
    on_message callback, then just sleep) decodes two cycles with NO explicit poll(): the constructor opens the audio source and
    starts the manage_cycle stand-in.  Audio = an iterator of 480-sample hops that advances a virtual clock (time_source / sleep
    are the reference's time_utils seam); it starts 3 hops late so the wall-clock re-sync at the grid wrap fires (the
    cycle that ends at the wrap must still be decoded)."""
    import threading
    import time as _t
    from pyft8_amd.receiver import Receiver
    a9, _, js9 = load_golden("test_09")
    a8, _, js8 = load_golden("test_08")
    vt = [0.0]
    lock = threading.Lock()
    got = []

    def hops():
        # two full cycles, then one more cycle of silence so that the second wrap (750 -> 0) is exercised too
        stream = np.concatenate([a9, a8, np.zeros(180000, np.int16)])
        for k in range(3 * 375):
            with lock:
                vt[0] = (k + 1) * 0.04 + 0.12                 # the stream lags the clock by 3 hops (PortAudio start-up latency)
            yield stream[480 * k:480 * k + 480]
            _t.sleep(0.0005)
        _t.sleep(0.5)

    def vsleep(dt):
        _t.sleep(0.002)

    rx = Receiver("any,keywords", got.append, sync_score_min=85, max_cands=200, time_source=lambda: vt[0], sleep=vsleep, audio_source=hops(),
                  early_decode_hop=None)
    try:
        assert rx._thread is not None and rx._thread.is_alive()
        deadline = _t.time() + 120
        while not getattr(rx.audio_in, "source_exhausted", False) and _t.time() < deadline:     # the CLI's `while True: sleep(1)`
            _t.sleep(0.05)
        _t.sleep(0.3)
    finally:
        rx.stop()
    assert rx.thread_error is None, rx.thread_error
    assert rx.audio_in.cycles_completed == 3
    txt = [" ".join(m["msg_tuple"]) for m in got]
    ref9 = [" ".join(m["msg_tuple"]) for m in js9["messages"]]
    ref8 = [" ".join(m["msg_tuple"]) for m in js8["messages"]]
    # cycle 1 complete and in order; cycle 2 the same decode set up to cross-cycle hash resolution (persistent call hashes)
    assert txt[:len(ref9)] == ref9
    second = txt[len(ref9):]
    assert len(second) >= len(ref8) - 1
    norm = lambda t: " ".join(w if not w.startswith("<") else "<>" for w in t.split(" "))      # noqa: E731
    assert len({norm(t) for t in second} & {norm(t) for t in ref8}) >= len(ref8) - 1
    assert {m["cyclestart_string"] for m in got[:len(ref9)]} == {"700101_000000"}
    assert {m["their_tx_cycle"] for m in got[len(ref9):]} == {1}


def test_streaming_early_decode_delivers_before_the_next_cycle():
    """VERDICT r2 #6 (reference receiver.py:389-401: candidates decode as their signals complete, first messages at ~12.9 s,
    tests/PyFT8.txt:1-19): the streaming receiver also decodes the partial cycle at hops 320 (12.8 s) and 340 (13.6 s) -- every signal
    whose payload symbols have arrived (start <= +0.96 s / +1.76 s), OSD decodes excepted -- and the rest at hop 375.  Under the virtual
    clock: the early passes' messages are delivered at 12.8 s and 13.6 s, they are a subset of the frame-complete decode set, together with the end-of-cycle pass they ARE that set (up
    to at most one OSD decode that depends on the padded tail), nothing is delivered twice, and a hop that arrives while the
    owner runs a two-pass decode_frames on the same Receiver does not disturb either (the live path has its own handle)."""
    from pyft8_amd.receiver import Receiver
    for name in ("test_09", "test_08"):
        audio, gold, js = load_golden(name)
        ref_txt = [" ".join(m["msg_tuple"]) for m in js["messages"]]
        vt = [0.0]
        got = []
        rx = Receiver("x", lambda d: got.append((vt[0], d)), time_source=lambda: vt[0])
        assert rx.early_decode_hops == (320, 340) and rx.early_decode_hop == 340
        side = None
        for k in range(375):
            vt[0] = (k + 1) * 0.04
            rx.audio_in._callback(audio[480 * k:480 * k + 480].tobytes(), 480, None, None)
            if k == 200:                      # the owner uses the same Receiver for a batch job in the middle of the cycle
                side = rx.decode_frames(np.stack([audio, audio]), passes=2)
            rx.poll()
        got = [(t, d) for t, d in got if "early" in d]           # (decode_frames delivers its own messages through on_message too)
        early = [(t, d) for t, d in got if d["early"]]
        late = [(t, d) for t, d in got if not d["early"]]
        assert early and all(min(abs(t - 12.8), abs(t - 13.6)) < 1e-9 for t, _ in early) and all(abs(t - 15.0) < 1e-9 for t, _ in late)
        first = [d for t, d in early if abs(t - 12.8) < 1e-9]
        # the 12.8-s pass: h0 <= 24 hops (+ a time tweak).  These recordings start ~0.9 s before their cycle (signals at +1.4 s), so only
        # test_09's three earliest signals qualify, none of test_08's; a receiver on the wall clock sees its signals at +0.5 s = hop 308
        assert all(d["tsec"] <= 1.0 for d in first) and len(first) == (3 if name == "test_09" else 0)
        e_txt = [" ".join(d["msg_tuple"]) for _, d in early]
        all_txt = e_txt + [" ".join(d["msg_tuple"]) for _, d in late]
        assert len(set(all_txt)) == len(all_txt)                                   # the per-cycle duplicate filter
        assert set(e_txt) <= set(ref_txt), sorted(set(e_txt) - set(ref_txt))       # early messages are frame-complete messages
        assert set(ref_txt) <= set(all_txt) and len(set(all_txt) - set(ref_txt)) <= 1, (sorted(set(all_txt) ^ set(ref_txt)))
        assert len(e_txt) >= 0.5 * len(ref_txt)                                    # most of the cycle arrives early
        assert all("OSD" not in d["decode_notes"] for _, d in early)
        assert all(d["tsec"] <= 1.8 for _, d in early)                             # h0 <= 44 hops (+ a time tweak)
        assert {d["cyclestart_string"] for _, d in got} == {"700101_000000"}
        # the mid-cycle batch job saw exactly what a fresh Receiver decodes, both frames alike
        fresh = Receiver("x", None).decode_frames(np.stack([audio, audio]), passes=2)
        assert [[" ".join(m["msg_tuple"]) for m in f] for f in side] == [[" ".join(m["msg_tuple"]) for m in f] for f in fresh]
        print(f"{name}: {len(first)} of {len(ref_txt)} messages delivered at 12.8 s, {len(e_txt) - len(first)} at 13.6 s, {len(late)} at 15.0 s")
        rx.stop()


def test_streaming_incremental_delivery_like_manage_cycle():
    """VERDICT r3 missing 4 (reference receiver.py:389-401: every candidate is decoded as soon as its payload has passed):
    early_decode_hop="incremental" decodes the partial cycle every 0.2 s from hop 300.  Under the virtual clock every early message is
    delivered within 0.2 s (+ one hop) of the arrival of its last payload symbol, never before it, nothing twice, and all passes
    together are the frame-complete set."""
    from pyft8_amd.receiver import Receiver
    audio, gold, js = load_golden("test_09")
    ref_txt = [" ".join(m["msg_tuple"]) for m in js["messages"]]
    vt = [0.0]
    got = []
    rx = Receiver("x", lambda d: got.append((vt[0], d)), time_source=lambda: vt[0], early_decode_hop="incremental")
    assert rx.early_decode_hops == tuple(range(300, 375, 5))
    for k in range(375):
        vt[0] = (k + 1) * 0.04
        rx.audio_in._callback(audio[480 * k:480 * k + 480].tobytes(), 480, None, None)
        rx.poll()
    rx.stop()
    txt = [" ".join(d["msg_tuple"]) for _, d in got]
    assert len(set(txt)) == len(txt) and set(ref_txt) <= set(txt) and len(set(txt) - set(ref_txt)) <= 1
    early = [(t, d) for t, d in got if d["early"]]
    assert len(early) >= 0.6 * len(ref_txt) and len({round(t, 6) for t, _ in early}) >= 3          # spread over several passes
    for t, d in early:
        h0 = int(round((d["tsec"] - (int(d["tweaks"].split("t:")[1].split()[0]) / 200.0 if "t:" in d["tweaks"] else 0.0)) * 25))
        t_payload = (h0 + 4 + 4 * 72 + 4) * 0.04                      # arrival of the hop after the last payload symbol
        assert t_payload - 1e-9 <= t <= max(t_payload, 12.0) + 0.2 + 0.04 + 1e-9, (t, t_payload, d["tsec"], d["tweaks"])      # (first pass: hop 300 = 12.0 s)
    # stop() released both GPU handles; the receiver re-creates what a later call needs
    assert rx._h is None and rx._live is None
    again = rx.decode_frames(audio[None])
    assert [" ".join(m["msg_tuple"]) for m in again[0]] == ref_txt and rx._h is not None
    rx.close()
    assert rx._h is None
    print(f"incremental: {len(early)} of {len(ref_txt)} messages before the end of the cycle, at " + ", ".join(f"{t:.1f}" for t in sorted({t for t, _ in early})) + " s")


def test_d2h_async_tickets():
    """ft8rx_d2h_async / _query / _event (include/ft8rx.h): device -> page-locked host copies on the handle's result-copy stream, polled
    by ticket -- what rank `dst` of the gather uses.  40 copies of device audio (more than the 32-event ring: the oldest tickets read as
    complete, the 33rd call waits for the first), every byte equal to the synchronous copy; a ticket that was never issued is an error."""
    import time
    from pyft8_amd import _lib
    h = _lib.Handle(max_frames=2)
    ptr = h.staging_ptr()
    h.synth_frames(ptr, 4242, 2, n_signals=5)
    want = h.download_audio(ptr, 2)
    n = want.nbytes
    bufs = [h.pinned_bytes(n) for _ in range(4)]
    tickets = []
    for i in range(40):
        bufs[i % 4][:] = 0
        t = h.d2h_async(bufs[i % 4].ctypes.data, ptr, n)
        tickets.append(t)
        t0 = time.perf_counter()
        while not h.d2h_done(t):
            assert time.perf_counter() - t0 < 10.0
        assert bufs[i % 4].tobytes() == want.tobytes(), i
        assert h.d2h_event(t)
    assert tickets == list(range(40)) and h.d2h_done(tickets[0]) and h.d2h_event(tickets[0]) is None      # reused since: long done, no event to hand out
    with pytest.raises(_lib.Ft8rxError):
        h.d2h_done(1000)
    h.close()


def test_error_paths_and_lifecycle():
    """C ABI error convention: negative return + ft8rx_last_error text, surfaced as Ft8rxError; never a crash."""
    from pyft8_amd import _lib
    with pytest.raises(_lib.Ft8rxError, match="configuration"):
        _lib.Handle(_lib.default_config(max_cands=5000))                    # > FT8RX_MAX_CANDS of the deep layouts (2048)
    with pytest.raises(_lib.Ft8rxError, match="configuration"):
        _lib.Handle(_lib.default_config(f0_lo=0))
    with pytest.raises(_lib.Ft8rxError, match="configuration"):
        _lib.Handle(_lib.default_config(h0_lo=_lib.MIN_H0 - 1))            # the middle Costas block would leave the fine-sync series
    with pytest.raises(_lib.Ft8rxError, match="configuration"):
        _lib.Handle(_lib.default_config(h0_lo=100, h0_hi=_lib.MAX_H0 + 1))
    with pytest.raises(_lib.Ft8rxError, match="device"):
        _lib.Handle(device=99)
    h = _lib.Handle(max_frames=2)
    with pytest.raises(_lib.Ft8rxError, match="n_frames"):
        h.decode_batch(np.zeros((3, 180000), np.int16))
    with pytest.raises(_lib.Ft8rxError):
        h.osd(np.zeros((1, 174), np.float32), 100, 2)
    rec, cnt, ev, evc = h.decode_batch(np.zeros((2, 180000), np.int16))      # still usable after errors
    assert list(cnt) == [0, 0]
    h.close()
    h.close()                                                              # idempotent
    from pyft8_amd.receiver import Receiver
    with pytest.raises(_lib.Ft8rxError, match="early_decode_hop"):
        Receiver("x", None, early_decode_hop=200)                          # before the payload of any signal can be complete
    for v, want in (((320, 340), (320, 340)), (340, (340,)), (None, ()), ([350, 310, 350], (310, 350))):
        rx = Receiver("x", None, early_decode_hop=v)
        assert rx.early_decode_hops == want and rx.early_decode_hop == (want[-1] if want else None)
        rx.stop()


def test_repeatable_across_runs_and_stream_counts(ocfg):
    """Same input => same records regardless of how many streams/chunks the batch is cut into; events may arrive in a
    different order (atomics) but package_frame orders them."""
    from pyft8_amd import _lib, synth, messages as M
    audio = synth.make_batch(81000, 16)
    outs = []
    for ns in (1, 2, 4, 1):
        h = _lib.Handle(max_frames=16)
        h.set_streams(ns)
        rec, cnt, ev, evc = h.decode_batch(audio)
        outs.append((rec.copy(), cnt.copy(), [[" ".join(m["msg_tuple"]) for m in M.package_frame(rec[f], int(cnt[f]), ev[f], int(evc[f]))]
                                              for f in range(16)]))
        h.close()
    for rec, cnt, msgs in outs[1:]:
        assert np.array_equal(cnt, outs[0][1]) and msgs == outs[0][2]
        for f in range(16):
            assert rec[f][:cnt[f]].tobytes() == outs[0][0][f][:cnt[f]].tobytes()


def test_subbatch_partition_invariance():
    """ft8rx_set_subbatch: a stream's share of a batch runs as consecutive sub-batches through the whole chain (cache residency for
    large batches).  Records and the (sorted) event log do not depend on the partition -- whole share, uneven tail, one frame per chain."""
    import hashlib
    from pyft8_amd import _lib
    B = 80
    h = _lib.Handle(max_frames=B)
    ptr = h.staging_ptr()
    h.synth_frames(ptr, 5200000, B, n_signals=40, snr_range=(-12.0, 8.0))

    def digest(res):
        rec, cnt, ev, evc = res
        hsh = hashlib.sha256()
        for f in range(B):
            hsh.update(rec[f, :cnt[f]].tobytes())
            hsh.update(np.sort(ev[f, :min(int(evc[f]), _lib.EVENT_CAP)], order=["cand", "ipass", "slot", "seq"]).tobytes())
        return hsh.hexdigest(), int(cnt.sum())
    outs = []
    for ns, sub in ((1, 0), (2, 0), (2, 24), (2, 128), (4, 7), (1, 1), (2, 16)):
        h.set_streams(ns)
        h.set_subbatch(sub)
        h.enqueue(ptr, B)
        h.enqueue(ptr, B)                                                 # free-running: two batches back to back
        h.fetch(B)
        outs.append(digest(h.fetch(B)))
    assert outs[0][1] > 100 * B // 2
    assert all(o == outs[0] for o in outs), outs
    with pytest.raises(_lib.Ft8rxError):
        h.set_subbatch(-1)
    h.close()


def test_decode_messages_single_call(H):
    """ft8rx_decode_messages (audio -> messages in one native call) == ft8rx_decode_batch + ft8rx_package_batch."""
    from pyft8_amd import _lib
    audio = np.stack([load_golden(n)[0] for n in ("test_09", "synth_100000", "test_08")])
    rec, cnt, ev, evc = H.decode_batch(audio)
    want, wc = _lib.package_batch(rec, cnt, ev, evc)
    got, gc = H.decode_messages(audio)
    assert np.array_equal(gc, wc) and gc.sum() > 40
    for f in range(3):
        assert got[f, :gc[f]].tobytes() == want[f, :wc[f]].tobytes()
    assert [b" ".join(m["f"]).decode() for m in got[0, :gc[0]]] == [" ".join(m["msg_tuple"]) for m in load_golden("test_09")[2]["messages"]]


def test_ladder_modes_give_the_same_records_and_messages():
    """ft8rx_set_ladder_mode: fine-stage BP in ladder order (three launches, default) vs all five AP variants in one launch --
    identical records and rendered messages; the one-launch event log is a superset (attempts the ladder would not have reached)."""
    from pyft8_amd import _lib, synth, messages as M
    audio = synth.make_batch(515000, 24)
    res = []
    for mode in (0, 1):
        h = _lib.Handle(max_frames=24)
        h.set_ladder_mode(mode)
        rec, cnt, ev, evc = h.decode_batch(audio)
        msgs, mcnt = _lib.package_batch(rec, cnt, ev, evc)
        py = [[" ".join(m["msg_tuple"]) for m in M.package_frame(rec[f], int(cnt[f]), ev[f], int(evc[f]))] for f in range(4)]
        res.append((rec.copy(), cnt.copy(), msgs.copy(), mcnt.copy(), py,
                    [set(map(tuple, ev[f, :min(int(evc[f]), _lib.EVENT_CAP)].tolist())) for f in range(24)]))
        h.close()
    a, b = res
    assert np.array_equal(a[1], b[1]) and np.array_equal(a[3], b[3]) and a[4] == b[4]
    n_extra = 0
    for f in range(24):
        assert a[0][f][:a[1][f]].tobytes() == b[0][f][:b[1][f]].tobytes()
        assert a[2][f][:a[3][f]].tobytes() == b[2][f][:b[3][f]].tobytes()
        assert a[5][f] <= b[5][f]
        n_extra += len(b[5][f] - a[5][f])
    assert int(a[3].sum()) > 100
    with pytest.raises(_lib.Ft8rxError):
        h2 = _lib.Handle(max_frames=1)
        try:
            h2.set_ladder_mode(2)
        finally:
            h2.close()


def test_full_size_batch_properties():
    """BASELINE config-2 size (4096 frames in one batch), checked through size-independent properties:
    batch-size independence (a frame decodes the same inside a 4096-frame batch as in a 64-frame one), independence of the
    stream/chunk count, a checksum over all records, frames without signals decode to nothing, and a truth-based yield floor."""
    import hashlib
    from pyft8_amd import _lib
    B = 4096
    h = _lib.Handle(max_frames=B)
    ptr = h.staging_ptr()
    truth = h.synth_frames(ptr, 9000000, B, n_signals=50, snr_range=(-10.0, 10.0))

    def digest(res, n):
        rec, cnt, ev, evc = res
        hsh = hashlib.sha256()
        for f in range(n):
            hsh.update(rec[f, :cnt[f]].tobytes())
            hsh.update(np.sort(ev[f, :min(int(evc[f]), _lib.EVENT_CAP)], order=["cand", "ipass", "slot", "seq"]).tobytes())
        return hsh.hexdigest()
    h.set_streams(4)
    h.enqueue(ptr, B)
    big = h.fetch(B)
    d4 = digest(big, B)
    h.set_streams(1)
    h.enqueue(ptr, B)
    assert digest(h.fetch(B), B) == d4                                   # chunking over streams changes nothing
    small = _lib.Handle(max_frames=64)
    for start in (0, 2048, 4032):                                         # the same frames in a small batch
        small.enqueue(ptr + start * _lib.NSAMP * 2, 64)
        r = small.fetch(64)
        assert digest(r, 64) == digest(tuple(a[start:start + 64] for a in big), 64), start
    small.close()
    msgs, mcnt = _lib.package_batch(*big)
    got = 0
    for f in range(0, B, 64):                                             # truth-based yield on a sample of frames
        want = {t["msg"] for t in truth[f]}
        got += len({b" ".join(m["f"]).decode() for m in msgs[f, :mcnt[f]]} & want)
    assert got / (B // 64) > 25                                           # >= 25 of 50 per frame (reference probe: 28-29)
    assert int(mcnt.min()) > 10 and int(mcnt.max()) < 60
    h.synth_frames(ptr, 9100000, 8, n_signals=0)                          # noise only
    h.enqueue(ptr, 8)
    rec, cnt, ev, evc = h.fetch(8)
    assert _lib.package_batch(rec, cnt, ev, evc)[1].sum() <= 1           # (an OSD false decode on pure noise is possible, not several)
    h.close()
    # BASELINE config 2 AS STATED: the same 4096 frames with LDPC BP 30 iterations + OSD depth 2
    kw = dict(bp_iters_b=30, osd_single=30, osd_double=2)
    h2 = _lib.Handle(_lib.default_config(**kw), max_frames=B)
    ptr2 = h2.staging_ptr()
    h2.synth_frames(ptr2, 9000000, B, n_signals=50, snr_range=(-10.0, 10.0))
    h2.set_streams(4)
    h2.enqueue(ptr2, B)
    big30 = h2.fetch(B)
    d30 = digest(big30, B)
    assert d30 != d4                                                      # ten more iterations do change some outcomes
    h2.set_streams(1)
    h2.enqueue(ptr2, B)
    assert digest(h2.fetch(B), B) == d30
    n30 = int(sum((big30[0][f][:big30[1][f]]["status"] == 1).sum() for f in range(B)))
    n20 = int(sum((big[0][f][:big[1][f]]["status"] == 1).sum() for f in range(B)))
    assert n30 >= n20                                                     # (BP 30 decodes at ipass 4 what BP 20 left to OSD or lost)
    import oracle as O
    ocfg30 = O.default_config(**_lib.fft_plans(), **kw)
    sample = [0, 777, 1500, 2048, 3333, 4095]
    audio30 = {f: h2.download_audio(ptr2 + f * _lib.NSAMP * 2, 1)[0] for f in sample}
    from pyft8_amd import messages as M
    for f in sample:                                                      # six frames against the oracle run with the same knobs
        r = O.decode_frame(audio30[f], ocfg30)
        rec, cnt, ev, evc = (a[f] for a in big30)
        assert int(cnt) == len(r["cands"])
        for i, c in enumerate(r["cands"]):
            g = rec[i]
            assert (int(g["status"]), int(g["f0_idx"]), int(g["h0_idx"])) == (c.status, c.f0_idx, c.h0_idx), (f, i)
            if c.status == 1:
                assert (int(g["ipass"]), int(g["ap"]), int(g["method"]), int(g["n_its"]), int(g["msg_lo"]), int(g["msg_hi"])) == \
                       (c.ipass, c.ap, c.method, c.n_its, c.msg_lo, c.msg_hi), (f, i)
        assert [" ".join(m["msg_tuple"]) for m in M.package_frame(rec, int(cnt), ev, int(evc))] == [" ".join(m["msg_tuple"]) for m in r["msgs"]], f
    h2.close()


def test_randomised_parity_sweep():
    """the randomised GPU-vs-oracle sweep (tools/parity_sweep.py) as a driver-run test: 96 never-seen frames with random
    recipes (0..70 signals, -24..+25 dB) over six kwargs / knob sets -- defaults, a tighter threshold, a narrow window, the wide
    build (to 5900 Hz), the extension knobs (BP 30, OSD 40/4), order-3 OSD with the distance gate -- random stream counts and ladder
    modes: every candidate record, every natively rendered message and every Python-rendered message identical to the oracle."""
    import sys
    from conftest import ROOT
    if os.path.join(ROOT, "tools") not in sys.path:
        sys.path.insert(0, os.path.join(ROOT, "tools"))              # a real import: the sweep's worker processes unpickle its functions by module name
    import parity_sweep as ps
    kws = [ps.KW[i] for i in (0, 2, 3, 4, 6, 7)]
    tot = ps.run_sweep(nb=6, fpb=16, seed=20261002, kw_list=kws, first_index=9500000, verbose=False)
    assert tot["frames"] == 96 and tot["kwargs_sets"] == 6 and tot["cands"] > 3000 and tot["msgs"] > 500, tot
    assert tot["bad"] == 0, tot
    print(f"parity sweep: {tot['frames']} frames, {tot['cands']} records, {tot['msgs']} messages identical to the oracle in {tot['seconds']:.0f} s")


def test_bench_contract_line():
    """bench.py must print exactly one JSON line with the driver's keys plus roofline and cpu_baseline."""
    import json as _json
    import subprocess
    import sys as _sys
    from conftest import ROOT
    out = subprocess.run([_sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "2", "--warmup", "1", "--frames", "32"],
                         capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.strip().startswith("{")]
    assert len(lines) == 1
    d = _json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 2 and d["warmup"] == 1 and d["higher_is_better"] is True
    assert d["vs_baseline"] is None and d["scaling"] == "weak" and d["unit"] == "frames/s" and "workload" in d["config"]
    # the line says what was measured: `value` = audio resident in HBM, the same batch every step; the SURVEY 8d
    # metric (page-locked host audio -> H2D -> kernels -> D2H -> messages) over the SAME number of steps under its own name
    assert "resident in HBM" in d["config"]["workload"] and "same batch every step" in d["config"]["workload"]
    assert "resident in HBM" in d["value_definition"]
    assert d["value_8d_host_audio"] > 1000 and d["value_8d_steps"] == d["steps"]
    r = d["roofline"]
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12 and r["peak"] == 8000.0
    assert "counter_frac" in r and (r["counter_frac"] is None or (r["traffic"] is not None and 0 < r["counter_frac"] < 1
                                                                 and abs(r["counter_frac"] - r["traffic"] / (r["kernel_ms"] * 1e-3) / 1e9 / r["peak"]) < 1e-9))
    c = d["cpu_baseline"]
    assert c["kind"] == "port" and c["cores"] == 1 and c["value"] > 0 and c["unit"] == "frames/s"
    m = c["decode_set_match"]                 # BASELINE's metric: "... decode-set match vs CPU ref" -- on the frames the baseline decodes anyway
    assert m["frames_compared"] > 0 and m["frames_identical"] == m["frames_compared"] and m["messages"] > 0
    assert d["value"] > 1000
