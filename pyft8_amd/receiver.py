"""Drop-in surface of PyFT8/receiver.py for the MI355X build: batched, frame-complete decoding.

    Receiver(input_device_keywords, on_message, sync_score_min=85, max_cands=200,
             search_freq_range=[100,3000], search_time_range=[-2.0,3.0], verbose=False)   reference receiver.py:310-336
      .audio_in.search_grid / .waterfall_data / .get_cycle_spectrum()                    reference receiver.py:225-306
      .search(cyclestart_string, odd_even, search_f_idxs) -> [Candidate]                  reference receiver.py:338-367
      .set_band(band)                                                                     reference receiver.py:369
      .decode_frames(audio_i16[B,180000]) -> list[list[message dict]]                     (batch entry)
    decode_frames(audio, **receiver_kwargs)                                               module-level convenience

The reference is a real-time, one-frame-per-15-s receiver driven by PortAudio; this build decodes whole
frames ("frame-complete" semantics) in batches on the GPU.  Messages are delivered through
`on_message(dict)` with the reference's keys, on the caller's thread, in the reference's emit order.
There is no CPU path: without libft8rx.so and an MI355X the constructor raises.

Live use: like the reference, the constructor starts the receiver when an audio source exists -- PortAudio through
PyAudio if that module is importable (reference receiver.py:252, 266-270), or any iterator of 480-sample int16 hops passed as
`audio_source` -- plus a daemon thread that stands in for Receiver.manage_cycle (receiver.py:336, 372-412): it wakes every 0.1 s on
the `time_source` / `sleep` seam (time_utils.py:7-14) and decodes every completed cycle (poll()).  `Receiver.start()` / `.stop()`
do the same explicitly.
"""
import threading as _threading
import time as _time

import numpy as np

from . import _lib
from . import messages as _m

T_CYC, SYM_RATE, SAMP_RATE = 15, 6.25, 12000
WATERFALL_DOWNSAMPLE = 2


def config_from_kwargs(sync_score_min=85, max_cands=200, search_freq_range=(100, 3000), search_time_range=(-2.0, 3.0), **ext):
    """Receiver kwargs -> ft8rx_config (index arithmetic of reference receiver.py:233-235, 319)."""
    df = SYM_RATE / 2
    cfg = _lib.default_config(
        sync_score_min=float(sync_score_min), max_cands=int(min(max_cands, 1 << 30)),
        f0_lo=int(search_freq_range[0] / df), f0_hi=int(search_freq_range[1] / df),
        h0_lo=int((search_time_range[0] + 0.5) * 4 * SYM_RATE), h0_hi=int((search_time_range[1] + 0.5) * 4 * SYM_RATE))
    # The reference takes any value here (receiver.py:311-313, 319, 366-367); this build has compile-time layouts.  Say which
    # kwarg is out of range instead of letting ft8rx_create answer "configuration out of the supported range" (include/ft8rx.h).
    if cfg.max_cands < 1:
        raise _lib.Ft8rxError(f"max_cands={max_cands}: at least one candidate per frame")
    # max_cands: no limit.  Receiver.search makes at most one candidate per f0 bin (receiver.py:341-365), so any value beyond the
    # number of bins of the search range keeps exactly the same list; more than 256 select the build with the deep candidate
    # layouts (libft8rx_wide.so: FT8RX_MAX_CANDS = 2048 > 1884 bins of the widest range).
    cfg.max_cands = min(cfg.max_cands, max(1, cfg.f0_hi - cfg.f0_lo))
    # search_time_range: any window the reference itself can search.  Its search reads grid rows h0 + 148 .. h0 + 172 of a 750-row grid
    # (receiver.py:322, 346-347; negative rows wrap, rows >= 750 are an IndexError), i.e. h0 in [-898, 577]; candidates whose middle
    # Costas block leaves the fine-sync series (h0 outside [-140, 220]: beyond -6.1 .. +8.3 s) are scored in the time domain with
    # clamped reads, as the reference does (kernels/fine_sync.hpp: k_fine_td).
    if cfg.h0_lo < _lib.MIN_H0 or cfg.h0_hi > _lib.MAX_H0 or cfg.h0_hi <= cfg.h0_lo:
        raise _lib.Ft8rxError(f"search_time_range={list(search_time_range)}: a non-empty window inside [{_lib.MIN_H0 / 25 - 0.5:.1f}, "
                              f"{_lib.MAX_H0 / 25 - 0.5:.2f}] s (FT8RX_MIN_H0 / FT8RX_MAX_H0: beyond it the reference's own search indexes "
                              "outside its 750-row grid, receiver.py:346-347; its default is [-2, 3])")
    if cfg.f0_lo < 4 or cfg.f0_hi > _lib.MAX_F0_WIDE or cfg.f0_lo >= cfg.f0_hi:
        raise _lib.Ft8rxError(f"search_freq_range={list(search_freq_range)}: supported are 12.5 .. {_lib.MAX_F0_WIDE * df:.0f} Hz, low < high "
                              "(above 3000 Hz the wide build libft8rx_wide.so is used; the reference fails beyond ~5940 Hz, receiver.py:181-182)")
    known = {f[0] for f in _lib.Config._fields_}
    for k, v in ext.items():          # extension knobs: bp_iters_b, osd_single, osd_double, osd_triple, osd_max_hd, ...
        if k not in known:            # like the reference's fixed signature (e.g. the CLI's misspelt `search_timerange`, pyft8.py:137)
            raise TypeError(f"Receiver() got an unexpected keyword argument '{k}'")
        setattr(cfg, k, v)
    return cfg


def frames_from_ragged(frames):
    """Ragged input -> int16 [B, 180000].  `frames` is an array [B, n] / [n] or a list of 1-D arrays, each at most 15 s long;
    shorter frames are padded with digital silence at the end (the reference's ring buffer is zero before the first hop,
    receiver.py:248; in frame-complete semantics the frame starts at cycle time 0, so the missing tail is what is silent).
    Longer input is an error: cut a recording into cycles yourself -- the cycle boundary is yours to choose."""
    if isinstance(frames, np.ndarray) and frames.ndim == 1:
        frames = [frames]
    out = np.zeros((len(frames), _lib.NSAMP), np.int16)
    for i, f in enumerate(frames):
        f = np.asarray(f)
        if f.ndim != 1:
            raise _lib.Ft8rxError(f"frame {i}: expected a 1-D array of int16 samples, got shape {f.shape}")
        if f.dtype != np.int16:
            if not np.issubdtype(f.dtype, np.integer):
                raise _lib.Ft8rxError(f"frame {i}: samples must be integers in int16 range (got {f.dtype})")
            if len(f) and (f.min() < -32768 or f.max() > 32767):
                raise _lib.Ft8rxError(f"frame {i}: samples outside the int16 range")
        if len(f) > _lib.NSAMP:
            raise _lib.Ft8rxError(f"frame {i}: {len(f)} samples > {_lib.NSAMP} (15 s at 12 kHz)")
        out[i, :len(f)] = f
    return out


# AP masks of the reference (receiver.py:21-27), as data: (name, first bit, forced bit values); 'CQ' also forces bits 74, 75 -> 0,
# 76 -> 1 and 57, 58 -> 0 (:113-116).  The 'RR73' mask is the reference's, quirk included.
AP_PATTERNS = (("NoAP", 0, ""), ("CQ", 0, "00000000000000000000000000100"), ("RR73", 58, "0111111001110101001"),
               ("73", 58, "0111111010010100001"), ("RRR", 58, "0111111010010010001"))
# the ipass ladder (receiver.py:68-107): step -> (attempt, AP variants, arguments)
_LADDER = {0: ("good91+ldpc", range(5), (35, 5, False)), 2: ("good91", range(2), None), 3: ("ldpc", range(2), (35, 5, False)),
           4: ("ldpc", range(5), (90, 20, True)), 5: ("osd", range(5), None)}


def frames_from_wav(path, cycle_offset_s=0.0):
    """A mono 16-bit 12 kHz .wav (what the reference's pipeline scripts read with the `wave` module, tests/pipeline/*.py; its two
    fixture recordings) -> int16 [n_frames, 180000]: the recording cut into consecutive 15-s cycles from `cycle_offset_s` on, the
    last one padded with digital silence (frames_from_ragged).  Anything else than mono / 16 bit / 12 kHz is rejected -- resampling
    is not this package's business."""
    import wave
    with wave.open(path, "rb") as w:
        if (w.getnchannels(), w.getsampwidth(), w.getframerate()) != (1, 2, SAMP_RATE):
            raise _lib.Ft8rxError(f"{path}: need mono 16-bit {SAMP_RATE} Hz, got {w.getnchannels()} ch / {8 * w.getsampwidth()} bit / {w.getframerate()} Hz")
        x = np.frombuffer(w.readframes(w.getnframes()), dtype="<i2")
    x = x[int(round(cycle_offset_s * SAMP_RATE)):]
    if len(x) == 0:
        return np.zeros((0, _lib.NSAMP), np.int16)
    return frames_from_ragged([x[i:i + _lib.NSAMP] for i in range(0, len(x), _lib.NSAMP)])


class Candidate:
    """One sync candidate (reference receiver.py:29-135): `origin`, `search_grid_bounds`, and -- for candidates returned by
    Receiver.search -- the reference's per-candidate state machine: decode(current_max_ipass) advances ONE ladder step per call
    (:68-107) and check_and_package(duplicate_filter) emits the message dict (:51-66).  Every step runs on the GPU through the
    stage entry points (ft8rx_llr_grid, ft8rx_fine, ft8rx_ldpc, ft8rx_osd, ft8rx_crc_valid); batched decoding
    (Receiver.decode_frames) does not go through this class -- it exists so that code written against the reference's
    Candidate API (the manage_cycle loop, tests/pipeline scripts) runs unchanged."""

    def __init__(self, origin, search_grid_bounds, record=None, rx=None, grid=None, llr_sd_min=5):
        self.origin = origin
        self.search_grid_bounds = search_grid_bounds
        self.record = record
        self.decode_result = None
        self._rx, self._grid = rx, grid
        self.on_message = rx.on_message if rx is not None else None
        self.llr_sd, self.llr_sd_min = 0, llr_sd_min
        self.ipass = 0
        self.source = None
        self.tweaks = "t:%+03d f:%+03d" % (0, 0)
        self.saved_llrs = []
        self.n_sync_matches = 100
        self.decode_notes = ""
        self.snr = 0

    # ---- receiver.py:51-66
    def check_and_package(self, duplicate_filter):
        self.msg_text = " ".join(self.decode_result)
        o = self.origin
        key = o["cyclestart_string"] + self.msg_text
        if key not in duplicate_filter:
            duplicate_filter.add(key)
            snr = "%+03d" % self.snr
            message = {"band": o["band"], "tsec": o["tsec"], "fHz": o["fHz"], "msg_tuple": self.decode_result, "their_snr": snr,
                       "their_tx_cycle": o["odd_even"],
                       "all_txt_format": f"{o['cyclestart_string']} {snr} {(o['tsec'] - 0.6):4.1f} {o['fHz']:4.0f} ~ {self.msg_text}",
                       "cyclestart_string": o["cyclestart_string"], "decode_completed": self._rx.time_source() if self._rx else _time.time(),
                       "tweaks": self.tweaks, "decode_notes": self.decode_notes + self.tweaks}
            if self.on_message is not None:
                self.on_message(message)
        self.decode_result = "stop"

    def _take_llr(self, llr, sd, snr, source):                       # the tail of _dB_to_llr (receiver.py:208-222)
        self.llr, self.llr_sd, self.snr, self.source = np.array(llr, np.float32), float(sd), int(snr), source
        if self.llr_sd <= self.llr_sd_min:
            self.decode_result = "stop"

    def _set_AP(self, k):                                            # receiver.py:109-117
        self.pat_name, b0, bits = AP_PATTERNS[k]
        self.llr = self.llr0.copy()
        for i, c in enumerate(bits):
            self.llr[b0 + i] = 5.0 if c == "1" else -5.0
        if self.pat_name == "CQ":
            self.llr[74:76] = -5.0
            self.llr[76] = 5.0
            self.llr[57:59] = -5.0

    def _attempt(self, what, args):                                  # _decode_good91 / _decode_ldpc / _decode_osd (receiver.py:119-135)
        from . import decoders as D
        if self.decode_result:
            return
        if what == "good91":
            self.decode_notes = f"{self.source}_{self.pat_name}_GOOD91 "
            self.decode_result = D.crc_unpack91(self.llr[:91])
        elif what == "ldpc":
            max_nc0, max_its, save = args
            self.decode_notes = f"{self.source}_{self.pat_name}_LDPC{max_its}"
            self.decode_result, self.n_its, out = D.ldpc_decode(self.llr, max_nc0, max_its)
            if save and not self.decode_result and len(out) == 174:
                self.saved_llrs.append((f"{self.pat_name}_LDPC{max_its}", out))
        else:
            self.decode_notes = f"{self.source}_{self.pat_name}_OSD"
            self.decode_result = D.osd_012(self.llr)

    def decode(self, current_max_ipass):
        """One step of the ipass ladder per call (reference receiver.py:68-107)."""
        if self._rx is None:
            raise _lib.Ft8rxError("this Candidate is a plain record view; Receiver.search() returns decodable ones")
        if self.ipass > current_max_ipass or self.decode_result == "stop":
            return
        rx, o = self._rx, self.origin
        step = self.ipass
        if step == 0:
            with rx._hlock:
                llr, sd, snr = rx._handle(1).llr_grid(self._grid, [0], [o["f0_idx"]], [o["h0_idx"]])
            self._take_llr(llr[0], sd[0], snr[0], "grid")
            self.llr0 = self.llr.copy()
        elif step == 1:                                              # _get_llr_fine (receiver.py:140-173)
            spec = rx.audio_in.get_cycle_spectrum()
            with rx._hlock:
                f = rx._handle(1).fine(spec[None], [0], [o["f0_idx"]], [o["h0_idx"]])
            tt, ft = int(f["ttweak"][0]), int(f["ftweak"][0])
            self.tweaks = " t:%+03d f:%+03d" % (tt, ft)
            self.n_sync_matches = int(f["nsync"][0])
            if self.n_sync_matches > 6:
                o.update({"tsec": float(o["tsec"] + tt / 200), "fHz": float(o["fHz"] + ft / 16)})
                self._take_llr(f["llr"][0], f["sd"][0], f["snr"][0], "fine")
            else:
                self.source = "fine"
                self.decode_result = "stop"
        elif step == 6:
            for self.pat_name, self.llr in self.saved_llrs:
                self._attempt("osd", None)
        elif step == 7:
            self.decode_result = "stop"
        if step == 2:
            self.llr0 = self.llr.copy()
        if step in _LADDER:
            what, variants, args = _LADDER[step]
            for k in variants:
                self._set_AP(k)
                if what == "good91+ldpc":
                    self._attempt("good91", None)
                    self._attempt("ldpc", args)
                else:
                    self._attempt(what, args)
        self.ipass += 1


def _pyaudio_device(audio_in, keywords):
    """True iff PyAudio is importable and a capture device matches `keywords` (sets audio_in.input_device_idx like the reference)."""
    try:
        import pyaudio
    except ImportError:
        return False
    try:
        return audio_in._find_input_device(keywords, pyaudio.PyAudio()) is not None
    except Exception:       # noqa: BLE001 -- no usable PortAudio host: behave like "no device"
        return False


def _as_frames(audio_i16):
    """[B, 180000] int16 as is; lists / short frames through frames_from_ragged."""
    if isinstance(audio_i16, np.ndarray) and audio_i16.dtype == np.int16 and audio_i16.ndim == 2 and audio_i16.shape[1] == _lib.NSAMP:
        return np.ascontiguousarray(audio_i16)
    if isinstance(audio_i16, np.ndarray) and audio_i16.ndim == 2:
        return frames_from_ragged(list(audio_i16))
    return frames_from_ragged(audio_i16)


class AudioIn:
    """Host-side state the reference keeps per receiver (receiver.py:225-306): the 15-s audio ring, the
    750 x 976 search grid that the GUI's waterfall views in place, and the hop pointer.

    Two ways to fill it:
      * load_frame(audio)  -- frame-complete: one GPU spectrogram launch for all 375 hops;
      * _callback(in_data, frame_count, time_info, status_flags) -- streaming: the PortAudio
        callback signature of the reference (receiver.py:295-306).  Each call appends one 480-sample hop,
        advances search_grid_ptr exactly like the reference (wall-clock re-sync at the grid wrap) and writes
        that hop's 976 dB values, computed on the GPU (ft8rx_hop_spectrum), into search_grid[ptr] in place.
    Unlike the reference, decoding is triggered when a cycle's 375 hops are complete (Receiver.poll), not at
    hop 260 -- the frame-complete semantics of this build."""

    def __init__(self, search_freq_range, receiver, input_device_keywords=None):
        self.input_device_idx = None
        self.input_device_keywords = input_device_keywords
        self.stream = None
        self._feeder = None
        self.search_hps, self.search_bpt = 4, 2
        self.search_freq_range = search_freq_range
        self.search_fft_len = 3840
        self.samples_perhop = 480
        self.df = SYM_RATE / self.search_bpt
        self.search_f0_idx_range = [int(search_freq_range[0] / self.df), int(search_freq_range[1] / self.df)]
        self.search_hops_per_cycle = int(T_CYC * SYM_RATE * self.search_hps)
        self.search_hops_per_grid = 2 * self.search_hops_per_cycle
        self.dt = T_CYC / self.search_hops_per_cycle
        self.samples_per_cycle = SAMP_RATE * T_CYC
        # reference receiver.py:240: f0_hi + 8 * bpt columns (976 at the default range)
        self.search_grid = np.ones((self.search_hops_per_grid, self.search_f0_idx_range[1] + 8 * self.search_bpt), dtype=np.float32)
        self._rx = receiver
        self.search_grid_ptr = int(self._grid_time() * self.search_hops_per_grid / (2 * T_CYC))
        self.audio_buffer = np.zeros(self.samples_per_cycle, dtype=np.int16)    # reference: float32 copies of int16 samples
        self._audio = None
        self.cycle_spectrum = None
        self.cycles_completed = 0
        self._ready = []                 # (int16 frame, cycle start time) of completed, not yet decoded cycles
        self._lock = _threading.Lock()   # _callback may run on an audio thread, poll() on the manage_cycle stand-in
        d = WATERFALL_DOWNSAMPLE
        self.waterfall_data = {"data": self.search_grid[::d, ::d].T, "df": self.df * d, "dt": self.dt * d,
                               "sig_w": int(79 * self.search_hps / d), "sig_h": int(8 * self.search_bpt / d),
                               "pixels_per_cycle": int(self.search_hops_per_cycle / d)}

    def _grid_time(self):
        return self._rx.time_source() % (2 * T_CYC)

    # ---- audio sources (reference receiver.py:252-270)
    def _find_input_device(self, input_device_keywords, pya):
        """First PortAudio device whose name contains every comma-separated keyword (reference receiver.py:254-264)."""
        for dev_idx in range(pya.get_device_count()):
            name = pya.get_device_info_by_index(dev_idx)["name"]
            if all(pat in name for pat in input_device_keywords.replace(" ", "").split(",")):
                self.input_device_idx = dev_idx
                break
        return self.input_device_idx

    def open(self, source=None):
        """Start feeding _callback.
        source None : PortAudio capture through PyAudio (imported lazily; reference receiver.py:266-270): int16 mono 12 kHz,
                      480 frames per buffer, this object's _callback as the stream callback;
        otherwise   : any iterable of hops -- 480 int16 samples as an array or bytes -- consumed by a daemon thread that calls
                      _callback per hop.  Pacing is the iterator's business (a live source blocks until the next hop exists; a
                      file replayer sleeps; a test advances its virtual clock inside the generator)."""
        if source is None:
            try:
                import pyaudio
            except ImportError as e:
                raise _lib.Ft8rxError("AudioIn.open(): PyAudio is not installed -- pass an iterator of 480-sample int16 hops instead") from e
            pya = pyaudio.PyAudio()
            if self.input_device_idx is None and self.input_device_keywords:
                self._find_input_device(self.input_device_keywords, pya)
            self.stream = pya.open(format=pyaudio.paInt16, channels=1, rate=SAMP_RATE, input=True, input_device_index=self.input_device_idx,
                                   frames_per_buffer=self.samples_perhop, stream_callback=self._pa_callback)
            self.stream.start_stream()
            return self

        def feed():
            try:
                for hop in source:
                    if self._rx._stop.is_set():
                        break
                    self._callback(hop if isinstance(hop, (bytes, bytearray, memoryview)) else np.ascontiguousarray(hop, np.int16).tobytes(),
                                   self.samples_perhop, None, None)
                else:
                    self.source_exhausted = True
            except Exception as e:          # noqa: BLE001 -- surfaced by Receiver.poll()/stop() instead of dying silently on a daemon thread
                self._rx.thread_error = e
            finally:
                self._feeder = None         # a later start() opens a source again instead of running with no input
        self.source_exhausted = False
        t = self._feeder = _threading.Thread(target=feed, name="ft8rx-audio-in", daemon=True)
        t.start()
        return self

    def _pa_callback(self, in_data, frame_count, time_info, status_flags):
        import pyaudio
        self._callback(in_data, frame_count, time_info, status_flags)
        return (None, pyaudio.paContinue)

    def close(self, timeout=5.0):
        """Stop the PortAudio stream / wait for the feeder thread (which ends at its next hop once the receiver's stop flag is set)."""
        if self.stream is not None:
            try:
                self.stream.stop_stream()
                self.stream.close()
            finally:
                self.stream = None
        t = self._feeder
        if t is not None and t is not _threading.current_thread():
            t.join(timeout)
            if not t.is_alive():
                self._feeder = None

    def load_frame(self, audio_i16):
        """Frame-complete stand-in for 375 calls of _callback (receiver.py:295-306)."""
        self._audio = np.ascontiguousarray(audio_i16, np.int16).reshape(_lib.NSAMP)
        self.audio_buffer[:] = self._audio
        with self._rx._hlock:
            g = self._rx._handle(1).spectrogram(self._audio)[0]
        self.search_grid[1:376] = g[1:376, :self.search_grid.shape[1]]          # in place: waterfall_data['data'] is a live view
        self.search_grid_ptr = 375
        self.cycle_spectrum = None

    def _callback(self, in_data, frame_count=None, time_info=None, status_flags=None):
        samples = np.frombuffer(in_data, dtype=np.int16)
        n = len(samples)
        self.audio_buffer[:-n] = self.audio_buffer[n:]
        self.audio_buffer[-n:] = samples
        self.search_grid_ptr = (self.search_grid_ptr + 1) % self.search_hops_per_grid
        # a cycle is complete when the pointer reaches a multiple of 375 -- decided BEFORE the wall-clock re-sync below, which
        # moves the pointer away from 0 whenever the stream lags the clock (normal after PortAudio's start-up latency)
        cycle_done = self.search_grid_ptr % self.search_hops_per_cycle == 0
        if self.search_grid_ptr == 0:
            tg = self._grid_time()
            if tg > 0.1:
                self.search_grid_ptr = int(tg * self.search_hops_per_grid / (2 * T_CYC))
        # the live path has a handle of its own (Receiver._live): a hop arriving while the owner runs decode_frames / search on the
        # receiver's main handle must not touch that handle's staging audio, scratch rows or result slots
        with self._rx._live_lock:
            self.search_grid[self.search_grid_ptr, :] = self._rx._live_handle().hop_spectrum(self.audio_buffer[-self.search_fft_len:])[:self.search_grid.shape[1]]
        in_cycle = self.search_grid_ptr % self.search_hops_per_cycle
        if cycle_done:                                                   # the last hop of a cycle just landed
            self._audio = self.audio_buffer.copy()
            self.cycle_spectrum = None
            self.cycles_completed += 1
            with self._lock:
                self._ready.append((self._audio, self._rx.time_source(), 0))
        elif in_cycle in self._rx.early_decode_hops:
            # early pass (reference receiver.py:389-401 decodes candidates as their signals complete, first messages at ~12.9 s): the
            # cycle so far, silence after it -- the payload of every signal that started by +0.96 s has arrived at hop 320 (12.8 s), by
            # +1.76 s at hop 340 (13.6 s)
            part = np.zeros(_lib.NSAMP, np.int16)
            n_have = in_cycle * self.samples_perhop
            part[:n_have] = self.audio_buffer[-n_have:]
            with self._lock:
                self._ready.append((part, self._rx.time_source(), in_cycle))
        return (None, 0)                 # (None, pyaudio.paContinue)

    def get_cycle_spectrum(self):
        """First 49152 (wide build: 96000) bins of the reference's 96001-bin spectrum (receiver.py:280-286): all the fine sync reads."""
        if self.cycle_spectrum is None:
            if self._audio is None:
                raise _lib.Ft8rxError("no frame loaded")
            with self._rx._hlock:
                self.cycle_spectrum = self._rx._handle(1).cycle_spectrum(self._audio)[0]
        return self.cycle_spectrum


class Receiver:
    def __init__(self, input_device_keywords, on_message, sync_score_min=85, max_cands=200,
                 search_freq_range=[100, 3000], search_time_range=[-2.5 + 0.5, 2.5 + 0.5], verbose=False,
                 device=0, max_frames=1, time_source=None, sleep=None, audio_source=None, autostart=None, early_decode_hop=(320, 340),
                 **extension_knobs):
        if search_freq_range[1] > 5900 or search_freq_range[0] < 12.5 or search_freq_range[0] >= search_freq_range[1]:
            # the reference sizes its grid from search_freq_range (receiver.py:234-240) and itself fails beyond ~5940 Hz, where the
            # fine-sync slice runs off the cycle spectrum (receiver.py:181-182).  Here the layouts are compile-time: up to 3000 Hz runs
            # on libft8rx.so, beyond that on libft8rx_wide.so (include/ft8rx.h), chosen by _lib.Handle from cfg.f0_hi
            raise _lib.Ft8rxError(f"search_freq_range {list(search_freq_range)} outside the supported [12.5, 5900] Hz")
        self.on_message = on_message
        self.time_source = time_source or _time.time          # the reference's time_utils seam (time_utils.py:7-8)
        self.sleep = sleep or _time.sleep                     # time_utils.py:13-14
        self._stop = _threading.Event()
        self._thread = None
        # Two handles, two locks.  The owner's calls (decode_frames, search, load_frame, get_cycle_spectrum) share the main handle
        # self._h under _hlock -- including its reallocation in _handle().  The live path (audio callback: hop spectra; manage_cycle
        # stand-in: poll) runs on its own one-frame handle self._live under _live_lock, so a hop that arrives in the middle of a
        # multi-pass decode cannot touch the main handle's staging audio, scratch rows or result slots.
        self._hlock = _threading.RLock()
        self._live_lock = _threading.RLock()
        self._live = None
        self.thread_error = None
        # streaming: also decode the partial cycle when these hops have arrived (320 = 12.8 s, 340 = 13.6 s into the cycle), so that
        # most messages are delivered before the next cycle starts, as the reference's are (its first decodes appear at ~12.9 s,
        # tests/PyFT8.txt:1-19); one hop, a sequence of hops, or None / 0 / () = only at the end of the cycle
        if isinstance(early_decode_hop, str):
            # "incremental": the reference starts on each candidate as soon as its payload has arrived (manage_cycle polls every 0.1 s from
            # hop 260, receiver.py:389-401); here the partial cycle is decoded every 5 hops (0.2 s) from hop 300 -- the payload of a signal
            # that starts at 0 s is complete at hop 296 -- so every message is delivered within 0.2 s of its last payload symbol
            if early_decode_hop != "incremental":
                raise _lib.Ft8rxError('early_decode_hop: a hop, a sequence of hops, None or "incremental"')
            early_decode_hop = tuple(range(300, 375, 5))
        if isinstance(early_decode_hop, range):
            early_decode_hop = tuple(early_decode_hop)
        hops = early_decode_hop if isinstance(early_decode_hop, (tuple, list)) else ((early_decode_hop,) if early_decode_hop else ())
        hops = tuple(sorted({int(x) for x in hops}))
        if any(not 300 <= x < 375 for x in hops):
            raise _lib.Ft8rxError("early_decode_hop: hops must lie in [300, 375) (the payload of a signal that starts at 0 s is complete at hop 296) or be None")
        self.early_decode_hops = hops
        self.early_decode_hop = hops[-1] if hops else None    # the last early pass
        self._cycle_seen = {}                                 # cycle start -> message texts already delivered (early pass, then the full one)
        self.sync_score_min, self.max_cands = sync_score_min, max_cands
        self.verbose = verbose
        self.band = None
        self.candidates = []
        self.cfg = config_from_kwargs(sync_score_min, max_cands, search_freq_range, search_time_range, **extension_knobs)
        self.search_h0_range = [self.cfg.h0_lo, self.cfg.h0_hi]
        self.search_start_hop = self.search_h0_range[1] + 43 * 4
        self.device = device
        self._h = None
        self.subtract_refine = 2              # multi-pass decode: how origins are re-estimated before subtraction (_lib.Handle.subtract):
                                              # 2 = on a decimated baseband copy (same yield as the full-rate scans of 1, 5x faster)
        self.call_hashes = _lib.CallHashTable()       # persistent across the cycles of the stream (poll); batches use fresh ones
        self._handle(max_frames)                      # fail loudly now if there is no GPU / library
        self.audio_in = AudioIn(search_freq_range, self, input_device_keywords)
        # The reference starts its audio thread and manage_cycle in the constructor (receiver.py:252, 336).  Same here when there
        # is something to listen to: an explicit audio_source, or PortAudio (PyAudio importable and a device matches the keywords).
        if autostart is None:
            autostart = audio_source is not None or (bool(input_device_keywords) and _pyaudio_device(self.audio_in, input_device_keywords))
        if autostart:
            self.start(audio_source)

    # ---- the manage_cycle stand-in (reference receiver.py:336, 372-412)
    def start(self, audio_source=None):
        """Open the audio source (AudioIn.open) and start the daemon that decodes completed cycles: every 0.1 s (the reference's
        poll period, receiver.py:383) it calls poll(); on_message runs on that thread, as in the reference.  Unlike the
        reference, whose daemon dies on the first exception, an exception is kept in .thread_error and the loop ends."""
        if self._thread is not None and self._thread.is_alive():
            return self
        self._stop.clear()
        self.thread_error = None
        if self.audio_in._feeder is not None and not self.audio_in._feeder.is_alive():
            self.audio_in._feeder = None
        if audio_source is not None or self.audio_in.stream is None and self.audio_in._feeder is None:
            self.audio_in.open(audio_source)

        def manage_cycle():
            try:
                while not self._stop.is_set():
                    self.sleep(0.1)
                    self.poll()
            except Exception as e:              # noqa: BLE001 -- surfaced to the owner instead of silently killing the daemon
                self.thread_error = e
        self._thread = _threading.Thread(target=manage_cycle, name="ft8rx-manage-cycle", daemon=True)
        self._thread.start()
        return self

    def stop(self, timeout=5.0):
        """Stop the daemon and the audio source, wait for both; an exception that ended either thread is raised here."""
        self._stop.set()
        if self._thread is not None and self._thread is not _threading.current_thread():
            self._thread.join(timeout)
        self.audio_in.close(timeout)
        self.close()
        self._raise_thread_error()

    def close(self):
        """Release the GPU handles (workspaces, page-locked buffers) of the batch and the live path; a later decode_frames / poll /
        search re-creates what it needs.  stop() calls this."""
        with self._hlock:
            if self._h is not None:
                self._h.close()
                self._h = None
        with self._live_lock:
            if self._live is not None:
                self._live.close()
                self._live = None

    def _raise_thread_error(self):
        e, self.thread_error = self.thread_error, None
        if e is not None:
            raise _lib.Ft8rxError(f"receiver thread failed: {type(e).__name__}: {e}") from e

    def _handle(self, n_frames):
        with self._hlock:
            if self._h is None or self._h.max_frames < n_frames:
                if self._h is not None:
                    self._h.close()
                self._h = _lib.Handle(self.cfg, device=self.device, max_frames=n_frames)
            # small batches: latency over work -- see ft8rx_set_ladder_mode
            self._h.set_ladder_mode(1 if n_frames < 128 else 0)       # crossover measured at 64..128 frames per call (profiles/archive/r02_latency.txt)
            return self._h

    def _live_handle(self):
        """The live path's own one-frame handle (hop spectra and the per-cycle decodes of poll); call with _live_lock held."""
        if self._live is None:
            self._live = _lib.Handle(self.cfg, device=self.device, max_frames=1)
            self._live.set_ladder_mode(1)
        return self._live

    def set_band(self, band):
        self.band = band

    def search(self, cyclestart_string, odd_even, search_f_idxs=None):
        """Costas sync search over one cycle of audio_in.search_grid (reference receiver.py:338-367).

        odd_even selects the half of the 750-row grid (rows odd_even*375 + 1 ... + 375); hops before the cycle's
        first row read 1.0 (frame-complete semantics, DESIGN.md section 1 -- the live reference would see the
        tail of the previous cycle there).  search_f_idxs: any sequence of f0 indices (the reference iterates over it, keeps
        the f0 whose best score clears the threshold, sorts by score -- stably, so ties keep the iteration order -- and cuts
        at max_cands).  The configured range runs entirely on the device (k_sync + k_topk, as in the decode pipeline); any other
        sequence is one pass of the correlation kernel over [min, max] of the list on the same handle (ft8rx_sync_scores) followed
        by those three steps on the host."""
        if odd_even not in (0, 1):
            raise _lib.Ft8rxError("odd_even must be 0 or 1")
        if search_f_idxs is None:
            idx = list(range(*self.audio_in.search_f0_idx_range))
        else:
            idx = [int(i) for i in search_f_idxs]
        if not idx:
            self.candidates = []
            return []
        width = self.audio_in.search_grid.shape[1]
        if min(idx) < 4 or max(idx) + 16 > width:
            raise _lib.Ft8rxError(f"search_f_idxs must stay within [4, {width - 16}] (the grid has {width} columns, receiver.py:240)")
        if len(set(idx)) != len(idx):
            raise _lib.Ft8rxError("search_f_idxs holds an index twice")
        r0 = odd_even * self.audio_in.search_hops_per_cycle
        rows = self.audio_in.search_grid[r0 + 1:r0 + 376] if odd_even == 0 else \
            np.concatenate([self.audio_in.search_grid[r0 + 1:], self.audio_in.search_grid[:1]])
        with self._hlock:
            h = self._handle(1)
            grid = np.ones((1, _lib.GRID_ROWS, h.grid_cols), np.float32)
            grid[0, 1:376, :width] = rows
            if idx == list(range(*self.audio_in.search_f0_idx_range)):
                # the configured range: threshold / stable sort / cut on the device (k_topk), as in the decode pipeline
                f0, h0, sc, cnt = h.sync_search(grid)
                found = [(int(f0[0, i]), int(h0[0, i]), float(sc[0, i])) for i in range(int(cnt[0]))]
            else:
                # any other index sequence: ONE pass of the same correlation kernel over [min, max] on this handle
                # (ft8rx_sync_scores), then the reference's own steps on the host -- keep the f0 whose best score clears the
                # threshold, in iteration order; sort by score, stably; cut at max_cands (receiver.py:350-367)
                lo, hi = min(idx), max(idx) + 1
                sc, h0 = h.sync_scores(grid, lo, hi)
                thr = np.float32(self.cfg.sync_score_min)
                found = [(f, int(h0[0, f - lo]), float(sc[0, f - lo])) for f in idx if sc[0, f - lo] > thr]
                found.sort(key=lambda c: -c[2])
                found = found[:self.cfg.max_cands]
        cands = []
        for f0i, h0i, sci in found:
            origin = {"h0_idx": h0i, "f0_idx": f0i, "tsec": h0i / 25.0, "fHz": 3.125 * f0i, "score": sci,
                      "cyclestart_string": cyclestart_string, "band": self.band, "odd_even": odd_even}
            cands.append(Candidate(origin, [r0 + h0i + 4, r0 + h0i + 4 * 71], rx=self, grid=grid))
        self.candidates = cands
        return cands

    def decode_frames(self, audio_i16, cyclestart_strings=None, return_records=False, passes=1, subtract_min_snr=-10,
                      sub_pass_osd=True, research="full"):
        """Decode B independent 15-s frames.  -> list (per frame) of message dicts in emit order.

        passes > 1 (extension): after each pass every newly decoded signal with SNR > subtract_min_snr is
        re-encoded and subtracted from the frame on the GPU (ft8rx_subtract -- the arithmetic of the reference experiment's
        Receiver.subtract_signal, tests/pipeline/receiver_sub.py:380-402, threshold :434) and the residual is decoded again;
        messages found that way are appended with "_SUB" added to decode_notes (as the reference tags them, :131-132).  Unlike
        the reference experiment -- which subtracts after every single decode, serially -- a pass subtracts all of a frame's
        new decodes at once, so whole batches stay on the GPU.  sub_pass_osd=False drops OSD decodes (ipass 5/6: first CRC-valid
        trial wins, the reference's source of false decodes) found in the later passes -- fewer false decodes, slightly less yield.
        research="local" is the experiment's re-search (receiver_sub.py:434-445), batched: the residual is searched only in the columns
        f0 - 2 .. f0 + 1 of the subtracted signals, with the sync threshold ignored (ft8rx_set_search_mask), and what that finds is not
        subtracted again -- one sweep for all of a frame's decodes where the experiment does one per decode."""
        audio = _as_frames(audio_i16)
        B = audio.shape[0]
        if B == 0:
            return ([], np.zeros((0, self.cfg.max_cands), _lib.RECORD_DTYPE), np.zeros(0, np.int32)) if return_records else []
        with self._hlock:
            return self._decode_frames_locked(audio, B, cyclestart_strings, return_records, passes, subtract_min_snr, sub_pass_osd, research)

    def _local_mask(self, msgs, mcnt, min_snr):
        """Search mask of the local re-search: columns f0 - 2 .. f0 + 1 (receiver_sub.py:440) of every message the sweep subtracts."""
        B = len(mcnt)
        lo, hi = self.cfg.f0_lo, self.cfg.f0_hi
        mask = np.zeros((B, hi - lo), np.uint8)
        sel = (np.arange(msgs.shape[1])[None, :] < np.asarray(mcnt)[:, None]) & (msgs["snr"] > min_snr)
        fi, mi = np.nonzero(sel)
        f0 = msgs["f0_idx"][fi, mi].astype(np.int64)
        for d in (-2, -1, 0, 1):
            c = f0 + d
            ok = (c >= lo) & (c < hi)
            mask[fi[ok], c[ok] - lo] = 1
        return mask

    def _residual_decode(self, h, B, local, msgs, mcnt, min_snr):
        """Decode the frames in the handle's staging buffer (after a subtraction sweep); local = the experiment's local re-search."""
        if not local:
            h.enqueue(h.staging_ptr(), B)
            return h.fetch(B)
        h.set_search_mask(self._local_mask(msgs, mcnt, min_snr))
        try:
            h.enqueue(h.staging_ptr(), B)
            return h.fetch(B)
        finally:
            h.set_search_mask(None)

    def _decode_frames_locked(self, audio, B, cyclestart_strings, return_records, passes, subtract_min_snr, sub_pass_osd, research="full"):
        if research not in ("full", "local"):
            raise _lib.Ft8rxError('research must be "full" or "local"')
        local = research == "local"
        h = self._handle(B)
        rec, cnt, ev, evc = h.decode_batch(audio)
        # host message layer: native, multithreaded (ft8rx_package_batch); messages.package_frame is its Python twin
        msgs, mcnt = _lib.package_batch(rec, cnt, ev, evc)
        cs = [cyclestart_strings[f] if cyclestart_strings is not None else "700101_000015" for f in range(B)]
        out = [_m.message_dicts(msgs[f], mcnt[f], cyclestart_string=cs[f], band=self.band, odd_even=0, on_message=self.on_message)
               for f in range(B)]
        seen = [{" ".join(d["msg_tuple"]) for d in out[f]} for f in range(B)]
        for _ in range(1, int(passes)):
            sigs = self._subtraction_list(msgs, mcnt, rec, subtract_min_snr)
            if not sigs[1].any():
                break
            h.subtract(h.staging_ptr(), B, sigs, refine=self.subtract_refine)    # decode_batch left the frames in the handle's device buffer
            rec, cnt, ev, evc = self._residual_decode(h, B, local, msgs, mcnt, subtract_min_snr)
            msgs, mcnt = _lib.package_batch(rec, cnt, ev, evc)
            keep = np.zeros(mcnt.shape, np.int32)
            for f in range(B):
                new = []
                for i, d in enumerate(_m.message_dicts(msgs[f], mcnt[f], cyclestart_string=cs[f], band=self.band, odd_even=0)):
                    t = " ".join(d["msg_tuple"])
                    if t not in seen[f] and (sub_pass_osd or "OSD" not in d["decode_notes"]):
                        seen[f].add(t)
                        d["decode_notes"] += "_SUB"
                        new.append(d)
                        msgs[f, keep[f]] = msgs[f, i]                # compact: only the new messages are subtracted next
                        keep[f] += 1
                        if self.on_message is not None:
                            self.on_message(d)
                out[f] += new
            mcnt = keep
            if local:
                break                                                # candidates of the local re-search are not subtracted again (:444)
        return (out, rec, cnt) if return_records else out

    @staticmethod
    def _subtraction_list(msgs, mcnt, rec, min_snr):
        """(signals[B, max] of _lib.SUBSIG_DTYPE, counts[B]): every emitted message with snr > min_snr, in emit order, with the
        tones of its codeword and the decoder's origin (fHz, tsec as the message dict reports them).  Native (ft8rx_subtraction_list);
        _subtraction_list_py is its numpy twin."""
        return _lib.subtraction_list(msgs, mcnt, rec, min_snr)

    @staticmethod
    def _subtraction_list_py(msgs, mcnt, rec, min_snr):
        B = len(mcnt)
        col = np.arange(msgs.shape[1])[None, :]
        sel = (col < np.asarray(mcnt)[:, None]) & (msgs["snr"] > min_snr)
        cnt = sel.sum(axis=1).astype(np.int32)
        ms = max(1, int(cnt.max()) if B else 1)
        arr = np.zeros((B, ms), _lib.SUBSIG_DTYPE)
        fi, mi = np.nonzero(sel)                                   # row-major: emit order within each frame
        if len(fi) == 0:
            return arr, cnt
        m = msgs[fi, mi]
        r = rec[fi, m["cand"]]
        slot = np.concatenate([np.arange(c) for c in cnt])
        fine = m["fine"] != 0
        arr["tones"][fi, slot] = _lib.encode_tones(r["msg_lo"], r["msg_hi"])
        arr["fHz"][fi, slot] = 3.125 * m["f0_idx"] + np.where(fine, m["ftweak"] / 16.0, 0.0)
        arr["tsec"][fi, slot] = m["h0_idx"] / 25.0 + np.where(fine, m["ttweak"] / 200.0, 0.0)
        return arr, cnt

    def decode_frames_arrays(self, audio_i16, n_threads=None, passes=1, subtract_min_snr=-10, sub_pass_osd=True, research="full"):
        """High-throughput variant of decode_frames: no Python dicts.  -> (messages[B, 128] of _lib.MESSAGE_DTYPE,
        counts[B], records[B, max_cands], record_counts[B]); rows are in the reference's emit order.  With passes > 1 (see
        decode_frames) the messages of the later passes are appended per frame and carry the pass index (1, 2, ...) in
        messages["pad"][:, :, 0]; the returned records are those of the first pass."""
        audio = _as_frames(audio_i16)
        B = audio.shape[0]
        if B == 0:
            raise _lib.Ft8rxError("empty batch")
        with self._hlock:
            return self._decode_frames_arrays_locked(audio, B, n_threads, passes, subtract_min_snr, sub_pass_osd, research)

    def _decode_frames_arrays_locked(self, audio, B, n_threads, passes, subtract_min_snr, sub_pass_osd, research="full"):
        if research not in ("full", "local"):
            raise _lib.Ft8rxError('research must be "full" or "local"')
        local = research == "local"
        h = self._handle(B)
        rec, cnt, ev, evc = h.decode_batch(audio)
        msgs, mcnt = _lib.package_batch(rec, cnt, ev, evc, n_threads=n_threads)
        if passes <= 1:
            return msgs, mcnt, rec, cnt
        out, ocnt = msgs.copy(), np.ascontiguousarray(mcnt, np.int32).copy()
        cur_m, cur_c, cur_rec = msgs, mcnt, rec
        for p in range(1, int(passes)):
            sigs = self._subtraction_list(cur_m, cur_c, cur_rec, subtract_min_snr)
            if not sigs[1].any():
                break
            h.subtract(h.staging_ptr(), B, sigs, refine=self.subtract_refine)
            rec2, cnt2, ev2, evc2 = self._residual_decode(h, B, local, cur_m, cur_c, subtract_min_snr)
            m2, c2 = _lib.package_batch(rec2, cnt2, ev2, evc2, n_threads=n_threads)
            new_m, new_c = _lib.merge_messages(out, ocnt, m2, c2, p, drop_osd=not sub_pass_osd)      # native: appends in place
            cur_m, cur_c, cur_rec = new_m, new_c, rec2
            if local:
                break
        return out, ocnt, rec, cnt

    def decode_frame(self, audio_i16, cyclestart_string="700101_000015"):
        return self.decode_frames(np.asarray(audio_i16)[None], [cyclestart_string])[0]

    # ---- streaming mode (stands in for the manage_cycle thread, receiver.py:372-412)
    def poll(self):
        """Decode every cycle that audio_in._callback has completed since the last call; messages go to on_message
        on the caller's thread.  -> list of message dicts.  Receiver.start() runs this on a daemon thread (the stand-in for
        manage_cycle); without it, call poll() from your own loop (e.g. every 0.1 s).

        Unlike batched decode_frames (independent frames, a fresh call-hash table per frame), the stream is ONE receiver: the
        frames share the persistent table self.call_hashes, as the reference's process-global databases.call_hashes does
        (databases.py:8), so a hashed / non-standard call heard in cycle N resolves `<...>` in cycle N+1."""
        self._raise_thread_error()                                     # an exception on the audio / daemon thread surfaces here
        out = []
        while True:
            with self.audio_in._lock:
                if not self.audio_in._ready:
                    break
                frame, t_now, early = self.audio_in._ready.pop(0)           # early: hops of the cycle received so far, 0 = the complete frame
            # start of the cycle the frame belongs to: a full frame is handed over at its end, an early one inside it
            t0 = T_CYC * int(t_now / T_CYC) if early else T_CYC * int((t_now - T_CYC / 2) / T_CYC)
            cs = _time.strftime("%y%m%d_%H%M%S", _time.gmtime(t0))
            with self._live_lock:
                rec, cnt, ev, evc = self._live_handle().decode_batch(frame[None])
            msgs, mcnt = _lib.package_batch(rec, cnt, ev, evc, n_threads=1, table=self.call_hashes)
            dicts = _m.message_dicts(msgs[0], mcnt[0], cyclestart_string=cs, band=self.band, odd_even=int((t0 % (2 * T_CYC)) / T_CYC))
            seen = self._cycle_seen.setdefault(t0, set())
            for k in [k for k in self._cycle_seen if k < t0 - 4 * T_CYC]:
                del self._cycle_seen[k]
            for i, d in enumerate(dicts):
                # early pass: only candidates whose PAYLOAD symbols (7..71) lie inside the hops received so far -- the reference's own
                # criterion for starting on a candidate (its search_grid_bounds, receiver.py:355,389; the last Costas block only
                # counts towards the sync gate) -- and no OSD decodes: first-CRC-valid-wins on a frame whose tail is padding
                # produces false decodes that the complete frame does not; what OSD finds is delivered by the end-of-cycle pass
                if early and (int(msgs[0, i]["h0_idx"]) + 4 + 4 * 72 + 4 > early or
                              int(msgs[0, i]["method"]) in (_lib.M_OSD, _lib.M_LDPC_B_OSD)):
                    continue
                text = " ".join(d["msg_tuple"])
                if text in seen:                                       # the reference's per-cycle duplicate filter (receiver.py:52-54)
                    continue
                seen.add(text)
                d["early"] = bool(early)
                out.append(d)
                if self.on_message is not None:
                    self.on_message(d)
        return out


def decode_frames(audio_i16, on_message=None, passes=1, research="full", **receiver_kwargs):
    """decode_frames(audio_i16[B,180000], **receiver_kwargs) -> list[list[message dict]]  (SURVEY.md 8b); passes > 1 adds the
    subtraction passes of Receiver.decode_frames."""
    audio = _as_frames(audio_i16)
    rx = Receiver("", on_message, max_frames=max(1, audio.shape[0]), **receiver_kwargs)
    return rx.decode_frames(audio, passes=passes, research=research)
