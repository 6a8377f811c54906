"""ctypes binding of the CPU oracle (oracle/ft8_oracle.c).  TEST INFRASTRUCTURE ONLY.

Importers allowed: tests/, __graft_entry__.smoke(), bench.py's cpu_baseline leg.
"""
import ctypes as C
import os
import subprocess
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "_build", "libft8oracle.so")
LIB_PATH_WIDE = os.path.join(HERE, "_build", "libft8oracle_wide.so")   # -DFT8O_WIDE: search_freq_range up to 5900 Hz (ft8_oracle.h)

NSAMP, GRID_ROWS, GRID_COLS, SPEC_BINS = 180000, 376, 976, 49152
GRID_COLS_WIDE, SPEC_BINS_WIDE, MAX_F0 = 1920, 96000, 960


class Config(C.Structure):
    _fields_ = [("sync_score_min", C.c_float), ("max_cands", C.c_int32),
                ("f0_lo", C.c_int32), ("f0_hi", C.c_int32), ("h0_lo", C.c_int32), ("h0_hi", C.c_int32),
                ("bp_nc0_a", C.c_int32), ("bp_iters_a", C.c_int32), ("bp_nc0_b", C.c_int32), ("bp_iters_b", C.c_int32),
                ("osd_single", C.c_int32), ("osd_double", C.c_int32), ("llr_sd_min", C.c_float),
                ("osd_triple", C.c_int32), ("osd_max_hd", C.c_int32),
                ("plan1920", C.c_int32 * 8), ("plan3200", C.c_int32 * 8), ("plan300", C.c_int32 * 8), ("plan320", C.c_int32 * 8)]


class Cand(C.Structure):
    _fields_ = [("f0_idx", C.c_int32), ("h0_idx", C.c_int32), ("score", C.c_float),
                ("grid_sd", C.c_float), ("fine_sd", C.c_float), ("snr_grid", C.c_int32), ("snr_fine", C.c_int32),
                ("ttweak", C.c_int32), ("ftweak", C.c_int32), ("nsync", C.c_int32),
                ("status", C.c_int32), ("ipass", C.c_int32), ("ap", C.c_int32), ("method", C.c_int32), ("n_its", C.c_int32),
                ("msg_lo", C.c_uint64), ("msg_hi", C.c_uint64)]


class Event(C.Structure):
    _fields_ = [("msg_lo", C.c_uint64), ("msg_hi", C.c_uint64), ("cand", C.c_int32), ("ipass", C.c_int32),
                ("valid", C.c_int32), ("pad", C.c_int32)]


class Msg(C.Structure):
    _fields_ = [("f", (C.c_char * 16) * 3), ("cand", C.c_int32), ("snr", C.c_int32), ("tsec", C.c_double), ("fHz", C.c_double),
                ("ipass", C.c_int32), ("ap", C.c_int32), ("method", C.c_int32), ("ttweak", C.c_int32), ("ftweak", C.c_int32),
                ("fine", C.c_int32)]


AP_NAMES = ["NoAP", "CQ", "RR73", "73", "RRR"]
_libs = {}


def _cpu_has_fma():
    try:
        with open("/proc/cpuinfo") as f:
            return any(" fma " in (ln.rstrip() + " ") for ln in f if ln.startswith("flags"))
    except OSError:
        return False


def build(force=False):
    src = max(os.path.getmtime(os.path.join(HERE, f)) for f in ("ft8_oracle.c", "ft8_oracle.h", "ft8_tables.h", "Makefile"))
    # the Makefile adds -mfma where the CPU has it (same results, faster fmaf); a library built on another machine with the flag must
    # not be loaded on a CPU without the instruction
    flags_file = os.path.join(HERE, "_build", ".flags")
    built_fma = False
    if os.path.exists(flags_file):
        with open(flags_file) as f:
            built_fma = "-mfma" in f.read()
    if built_fma and not _cpu_has_fma():
        force = True
    if force or any(not os.path.exists(p) or os.path.getmtime(p) < src for p in (LIB_PATH, LIB_PATH_WIDE)):
        subprocess.check_call(["make", "-s", "-B" if force else "-s", "-C", HERE])
    return LIB_PATH


def lib(wide=False):
    """The oracle library; wide=True -> the build with the wide layouts (same source, -DFT8O_WIDE).  The wrappers below pick the
    variant from their arguments: a config with f0_hi > 960, a 1920-column grid or a 96000-bin spectrum mean the wide one."""
    wide = bool(wide)
    if wide not in _libs:
        build()
        L = C.CDLL(LIB_PATH_WIDE if wide else LIB_PATH)
        L.ft8o_log10f.restype = C.c_float; L.ft8o_log10f.argtypes = [C.c_float]
        L.ft8o_tanhf.restype = C.c_float; L.ft8o_tanhf.argtypes = [C.c_float]
        L.ft8o_unpack77.argtypes = [C.c_void_p, C.c_uint64, C.c_uint64, C.c_void_p]
        L.ft8o_valid77.argtypes = [C.c_uint64, C.c_uint64]
        L.ft8o_hash_new.restype = C.c_void_p
        L.ft8o_hash_free.argtypes = [C.c_void_p]
        _libs[wide] = L
    return _libs[wide]


def _wide(cfg=None, grid=None, spec=None):
    return (cfg is not None and cfg.f0_hi > MAX_F0) or (grid is not None and grid.shape[-1] == GRID_COLS_WIDE) or \
        (spec is not None and spec.shape[-1] == SPEC_BINS_WIDE)


def default_config(**kw):
    c = Config()
    lib().ft8o_default_config(C.byref(c))
    for k, v in kw.items():
        if k.startswith("plan"):
            arr = getattr(c, k)
            for i in range(8):
                arr[i] = v[i] if i < len(v) else 0
        else:
            setattr(c, k, v)
    return c


def _p(a, t=C.c_float):
    return a.ctypes.data_as(C.POINTER(t))


def log10f(x):
    x = np.asarray(x, np.float32)
    L = lib()
    return np.array([L.ft8o_log10f(float(v)) for v in x.ravel()], np.float32).reshape(x.shape)


def tanhf(x):
    x = np.asarray(x, np.float32)
    L = lib()
    return np.array([L.ft8o_tanhf(float(v)) for v in x.ravel()], np.float32).reshape(x.shape)


def fft(x, plan):
    x = np.ascontiguousarray(x, np.complex64).copy()
    scr = np.empty_like(x)
    pl = (C.c_int32 * 8)(*(list(plan) + [0] * (8 - len(plan))))
    lib().ft8o_fft(_p(x.view(np.float32)), C.c_int(len(x)), pl, _p(scr.view(np.float32)))
    return x


def spectrogram(audio, cfg=None):
    cfg = cfg or default_config()
    audio = np.ascontiguousarray(audio, np.int16)
    g = np.empty((GRID_ROWS, GRID_COLS_WIDE if _wide(cfg) else GRID_COLS), np.float32)
    lib(_wide(cfg)).ft8o_spectrogram(_p(audio, C.c_int16), C.byref(cfg), _p(g))
    return g


def sync_search(grid, cfg=None):
    cfg = cfg or default_config()
    grid = np.ascontiguousarray(grid, np.float32)
    out = (Cand * 2048)()
    n = lib(_wide(cfg, grid)).ft8o_sync_search(_p(grid), C.byref(cfg), out)
    return [out[i] for i in range(n)]


def payload(grid, f0, h0):
    grid = np.ascontiguousarray(grid, np.float32)
    p = np.empty((58, 8), np.float32)
    lib(_wide(grid=grid)).ft8o_payload(_p(grid), int(f0), int(h0), _p(p))
    return p


def db_to_llr(p):
    p = np.ascontiguousarray(p, np.float32)
    llr = np.empty(174, np.float32)
    sd, snr = C.c_float(), C.c_int32()
    ok = lib().ft8o_db_to_llr(_p(p), _p(llr), C.byref(sd), C.byref(snr))
    return llr, sd.value, snr.value, bool(ok)


def cycle_spectrum(audio, cfg=None):
    cfg = cfg or default_config()
    audio = np.ascontiguousarray(audio, np.int16)
    s = np.empty(SPEC_BINS_WIDE if _wide(cfg) else SPEC_BINS, np.complex64)
    lib(_wide(cfg)).ft8o_cycle_spectrum(_p(audio, C.c_int16), C.byref(cfg), _p(s.view(np.float32)))
    return s


def fine(spec, f0, h0, cfg=None):
    cfg = cfg or default_config()
    spec = np.ascontiguousarray(spec, np.complex64)
    tt, ft, ns, snr = C.c_int32(), C.c_int32(), C.c_int32(), C.c_int32()
    sd = C.c_float()
    llr = np.zeros(174, np.float32)
    sg = np.zeros((79, 8), np.float32)
    r = lib(_wide(cfg, spec=spec)).ft8o_fine(_p(spec.view(np.float32)), C.byref(cfg), int(f0), int(h0), C.byref(tt), C.byref(ft), C.byref(ns),
                        _p(llr), C.byref(sd), C.byref(snr), _p(sg))
    return dict(ret=r, ttweak=tt.value, ftweak=ft.value, nsync=ns.value, llr=llr, sd=sd.value, snr=snr.value, sgrid=sg)


def set_ap(llr0, ap):
    llr0 = np.ascontiguousarray(llr0, np.float32)
    out = np.empty(174, np.float32)
    lib().ft8o_set_ap(_p(llr0), int(ap), _p(out))
    return out


def msg_int(lo, hi):
    return (int(hi) << 64) | int(lo)


def ldpc(llr, max_nc0, max_iters):
    """-> (ok, bits77:int|None, n_its, llr_out|None)"""
    llr = np.ascontiguousarray(llr, np.float32).copy()
    lo, hi = C.c_uint64(), C.c_uint64()
    nits, has = C.c_int32(), C.c_int32()
    ok = lib().ft8o_ldpc(_p(llr), int(max_nc0), int(max_iters), C.byref(lo), C.byref(hi), C.byref(nits), C.byref(has))
    return bool(ok), (msg_int(lo.value, hi.value) if ok else None), (nits.value if ok else -1), (llr if has.value else None)


def osd(llr, singles=30, doubles=2, triples=0, max_hd=0, want_hd=False):
    """osd_012 (decoders.py:223-272); triples / max_hd are the build's order-3 and acceptance-gate extensions (0 = reference)."""
    llr = np.ascontiguousarray(llr, np.float32)
    lo, hi = C.c_uint64(), C.c_uint64()
    trial, hd = C.c_int32(), C.c_int32(-1)
    cols = np.zeros(91, np.int32)
    ok = lib().ft8o_osd_ext(_p(llr), int(singles), int(doubles), int(triples), int(max_hd), C.byref(lo), C.byref(hi), C.byref(trial),
                            _p(cols, C.c_int32), C.byref(hd))
    out = (bool(ok), (msg_int(lo.value, hi.value) if ok else None), trial.value, cols)
    return out + (hd.value,) if want_hd else out


def argsort_f32(x):
    """np.argsort(x) of <= 256 float32 values as the reference's numpy (2.2.6 on AVX-512) orders them (ft8o_argsort_f32)."""
    x = np.ascontiguousarray(x, np.float32)
    out = np.zeros(len(x), np.int32)
    if lib().ft8o_argsort_f32(_p(x), len(x), _p(out, C.c_int32)) != 0:
        raise ValueError("ft8o_argsort_f32: 0 <= n <= 256")
    return out


def crc_valid91(llr91):
    llr91 = np.ascontiguousarray(llr91, np.float32)
    lo, hi = C.c_uint64(), C.c_uint64()
    r = lib().ft8o_crc_valid91(_p(llr91), C.byref(lo), C.byref(hi))
    return r, (msg_int(lo.value, hi.value) if r else None)


class HashTable:
    def __init__(self):
        self.h = C.c_void_p(lib().ft8o_hash_new())

    def unpack(self, bits77):
        buf = ((C.c_char * 16) * 3)()
        ok = lib().ft8o_unpack77(self.h, C.c_uint64(bits77 & (2 ** 64 - 1)), C.c_uint64(bits77 >> 64), C.byref(buf))
        return tuple(buf[i].value.decode() for i in range(3)) if ok else None

    def __del__(self):
        try:
            lib().ft8o_hash_free(self.h)
        except Exception:
            pass


def subtract(audio_f32, tones79, fHz, tsec):
    """In-place reference-style subtraction of one signal from a float32 180000-sample buffer; -> True if subtracted."""
    assert audio_f32.dtype == np.float32 and audio_f32.shape == (NSAMP,) and audio_f32.flags.c_contiguous
    t = np.ascontiguousarray(tones79, np.uint8)
    assert t.shape == (79,)
    f = lib().ft8o_subtract
    f.argtypes = [C.c_void_p, C.c_void_p, C.c_double, C.c_double]
    return bool(f(audio_f32.ctypes.data, t.ctypes.data, float(fHz), float(tsec)))


def valid77(bits77):
    return bool(lib().ft8o_valid77(C.c_uint64(bits77 & (2 ** 64 - 1)), C.c_uint64(bits77 >> 64)))


def decode_frame(audio, cfg=None):
    """Whole-frame oracle decode -> dict(cands, events, msgs)."""
    cfg = cfg or default_config()
    audio = np.ascontiguousarray(audio, np.int16)
    assert audio.shape == (NSAMP,)
    cands = (Cand * max(1, cfg.max_cands))()
    log = (Event * 4096)()
    mcap = max(256, cfg.max_cands)
    msgs = (Msg * mcap)()
    nc, nl, nm = C.c_int32(), C.c_int32(), C.c_int32()
    lib(_wide(cfg)).ft8o_decode_frame(_p(audio, C.c_int16), C.byref(cfg), cands, C.byref(nc), log, 4096, C.byref(nl), msgs, mcap, C.byref(nm))
    out_msgs = []
    for i in range(min(nm.value, mcap)):
        m = msgs[i]
        out_msgs.append(dict(msg_tuple=tuple(m.f[k].value.decode() for k in range(3)), cand=m.cand, snr=m.snr,
                             tsec=m.tsec, fHz=m.fHz, ipass=m.ipass, ap=m.ap, method=m.method,
                             ttweak=m.ttweak, ftweak=m.ftweak, fine=bool(m.fine)))
    return dict(cands=[cands[i] for i in range(nc.value)],
                events=[log[i] for i in range(min(nl.value, 4096))], n_events=nl.value, msgs=out_msgs)


def notes_of(m):
    """decode_notes + tweaks string exactly as the reference formats them (receiver.py:42,57,121,126,133,162)."""
    src = "fine" if m["fine"] else "grid"
    ap = AP_NAMES[m["ap"]]
    meth = {0: "GOOD91 ", 1: "LDPC5", 2: "LDPC20", 3: "OSD", 4: "LDPC20_OSD"}[m["method"]]
    tw = (" " if m["fine"] else "") + f"t:{m['ttweak']:+03d} f:{m['ftweak']:+03d}"
    return f"{src}_{ap}_{meth}" + tw


# ---- signal subtraction / multi-pass decode ---------------------------------------------------------------------
def encode_tones(bits77):
    """77-bit word -> uint8[79] tones (transmitter.py:181-223 encode_bits77)."""
    t = np.zeros(79, np.uint8)
    f = lib().ft8o_encode_tones
    f.argtypes = [C.c_uint64, C.c_uint64, C.c_void_p]
    f.restype = None
    f(C.c_uint64(bits77 & (2 ** 64 - 1)), C.c_uint64(bits77 >> 64), t.ctypes.data)
    return t


def cycle_spectrum_f32(audio_f32, cfg=None):
    cfg = cfg or default_config()
    a = np.ascontiguousarray(audio_f32, np.float32)
    assert a.shape == (NSAMP,)
    s = np.empty(SPEC_BINS_WIDE if _wide(cfg) else SPEC_BINS, np.complex64)
    lib(_wide(cfg)).ft8o_cycle_spectrum_f32(_p(a), C.byref(cfg), _p(s.view(np.float32)))
    return s


def refine_time_origin(audio_f32, fHz, tsec, cfg=None):
    """Candidate.refine_time_origin (receiver_sub.py:58-72) on the float32 residual -> (fHz, tsec, score)."""
    cfg = cfg or default_config()
    a = np.ascontiguousarray(audio_f32, np.float32)
    assert a.shape == (NSAMP,)
    f, t, sc = C.c_double(float(fHz)), C.c_double(float(tsec)), C.c_float()
    fn = lib(_wide(cfg)).ft8o_refine_time_origin
    fn.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    fn.restype = None
    fn(a.ctypes.data, C.byref(cfg), C.byref(f), C.byref(t), C.byref(sc))
    return f.value, t.value, sc.value


def refine2_subtract(audio_f32, tones79, fHz, tsec, subtract=True):
    """The build's refine = 2 (decimated baseband) for one signal, in place on a float32 buffer -> (subtracted, fHz, tsec)."""
    assert audio_f32.dtype == np.float32 and audio_f32.shape == (NSAMP,) and audio_f32.flags.c_contiguous
    t = np.ascontiguousarray(tones79, np.uint8)
    f, ts = C.c_double(float(fHz)), C.c_double(float(tsec))
    fn = lib().ft8o_refine2_subtract
    fn.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
    done = fn(audio_f32.ctypes.data, t.ctypes.data, C.byref(f), C.byref(ts), int(bool(subtract)))
    return bool(done), f.value, ts.value


def subtraction_list(res, min_snr=-10):
    """The signals a subtraction sweep removes (ft8rx_subtraction_list): every message of the frame with snr > min_snr, in emit order,
    as (tones, fHz, tsec) with the origin the message dict reports (receiver.py:166)."""
    out = []
    for m in res["msgs"]:
        if m["snr"] <= int(np.floor(min_snr)):
            continue
        c = res["cands"][m["cand"]]
        fHz = 3.125 * c.f0_idx + (m["ftweak"] / 16.0 if m["fine"] else 0.0)
        tsec = c.h0_idx / 25.0 + (m["ttweak"] / 200.0 if m["fine"] else 0.0)
        out.append((encode_tones(msg_int(c.msg_lo, c.msg_hi)), fHz, tsec))
    return out


def to_int16(audio_f32):
    """k_sub_to_i16: round half to even, saturate."""
    return np.clip(np.rint(audio_f32), -32768, 32767).astype(np.int16)


def set_search_mask(mask, cfg=None):
    """ft8o_set_search_mask: mask[f0_hi - f0_lo] of bytes (local re-search, receiver_sub.py:434-445) or None = the configured search;
    process-global in both oracle builds."""
    for wide in (False, True):
        try:
            L = lib(wide)
        except Exception:
            continue
        if mask is None:
            L.ft8o_set_search_mask(None, 0)
        else:
            m = np.ascontiguousarray(mask, np.uint8)
            L.ft8o_set_search_mask(_p(m, C.c_uint8), len(m))


def local_search_mask(sig_f0, cfg):
    """The columns the experiment re-searches after subtracting signals at the search columns sig_f0: range(f0 - 2, f0 + 2) each
    (receiver_sub.py:440), clipped to the configured range."""
    m = np.zeros(cfg.f0_hi - cfg.f0_lo, np.uint8)
    for f0 in sig_f0:
        for c in range(int(f0) - 2, int(f0) + 2):
            if cfg.f0_lo <= c < cfg.f0_hi:
                m[c - cfg.f0_lo] = 1
    return m


def decode_frame_passes(audio, cfg=None, passes=2, min_snr=-10, refine=2, drop_osd=False, research="full"):
    """Multi-pass decode of one frame as pyft8_amd.Receiver.decode_frames_arrays(passes=...) composes it (extension):
    decode; subtract every new message with snr > min_snr, in emit order, from a float32 copy (refine = 2: origin re-estimated on the
    decimated baseband copy; 3: the reference experiment's refine_time_origin then its subtract_signal; 0: subtract_signal as is);
    round to int16; decode the residual; append the messages whose text the frame does not have yet.  research = "local": the
    residual is searched only in the columns f0 - 2 .. f0 + 1 of the subtracted signals, with the sync threshold ignored, and what that
    finds is not subtracted again (the experiment's scheduling, receiver_sub.py:434-445, batched: one sweep instead of one per decode).
    -> dict(msgs = [(pass, msg dict)], origins = per sweep the refined (fHz, tsec) list, residual = int16 audio after the last sweep)"""
    cfg = cfg or default_config()
    audio = np.ascontiguousarray(audio, np.int16)
    res = decode_frame(audio, cfg)
    out = [(0, m) for m in res["msgs"]]
    fresh = res
    origins = []
    cur = audio
    for p in range(1, int(passes)):
        sigs = subtraction_list(fresh, min_snr)
        if not sigs:
            break
        sig_f0 = [fresh["cands"][m["cand"]].f0_idx for m in fresh["msgs"] if m["snr"] > int(np.floor(min_snr))]
        wf = cur.astype(np.float32)
        sweep = []
        for tones, fHz, tsec in sigs:
            if refine == 2:
                _, fHz, tsec = refine2_subtract(wf, tones, fHz, tsec, True)
            elif refine == 3:
                fHz, tsec, _ = refine_time_origin(wf, fHz, tsec, cfg)
                subtract(wf, tones, fHz, tsec)
            else:
                subtract(wf, tones, fHz, tsec)
            sweep.append((fHz, tsec))
        origins.append(sweep)
        cur = to_int16(wf)
        if research == "local":
            set_search_mask(local_search_mask(sig_f0, cfg))
            try:
                r2 = decode_frame(cur, cfg)
            finally:
                set_search_mask(None)
        else:
            r2 = decode_frame(cur, cfg)
        have = {m["msg_tuple"] for _, m in out}
        new = []
        for m in r2["msgs"]:
            if drop_osd and m["method"] in (3, 4):
                continue
            if m["msg_tuple"] in have:
                continue
            have.add(m["msg_tuple"])
            new.append(m)
            out.append((p, m))
        fresh = dict(cands=r2["cands"], msgs=new)
        if research == "local":
            break                                                   # candidates of the local re-search are not subtracted again (:444)
    return dict(msgs=out, origins=origins, residual=cur)
