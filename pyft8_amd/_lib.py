"""ctypes binding of libft8rx.so (include/ft8rx.h).  There is NO CPU fallback: if the HIP library is
missing or no MI355X is visible, the product raises."""
import ctypes as C
import os
import subprocess
import weakref

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("FT8RX_LIB", os.path.join(HERE, "libft8rx.so"))   # FT8RX_LIB: A/B builds of the same ABI
# the same source with the wide layouts (-DFT8RX_WIDE, include/ft8rx.h): search_freq_range up to 5900 Hz and up to 2048 candidates per
# frame (max_cands > 256); loaded only when a config asks for either
LIB_PATH_WIDE = os.environ.get("FT8RX_LIB_WIDE", os.path.join(HERE, "libft8rx_wide.so"))
SRC = os.path.join(HERE, "csrc", "ft8rx.hip")
# second translation unit: the FFT kernels (k_fine, k_spectrogram), compiled with the ILP scheduling strategy -- 3.9 % / 3 % faster for
# them, 56 % slower for k_bp, and the strategy is a per-translation-unit choice (csrc/ft8rx_ilp.hip, profiles/archive/r03_notes.md)
SRC_ILP = os.path.join(HERE, "csrc", "ft8rx_ilp.hip")
ILP_FLAGS = ["-mllvm", "-amdgpu-sched-strategy=iterative-ilp"]
# -fno-slp-vectorize: on gfx950 a v_pk_add/mul_f32 issues at exactly the cost of the two scalar ops it replaces (tools/ubench/valu_rate.hip,
# profiles/archive/r02_valu_rate.txt) while the packing costs ~1000 extra v_mov in k_fine: scalar code is 7 % faster there, bit-identical.
HIPCC_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fno-slp-vectorize", "-Wno-unused-result",
               "-Wno-unused-value", "-fPIC", "-shared"]

NSAMP, GRID_ROWS, GRID_COLS, SPEC_BINS, MAX_CANDS, EVENT_CAP = 180000, 376, 976, 49152, 256, 512
MIN_H0, MAX_H0 = -898, 578                          # FT8RX_MIN_H0 / FT8RX_MAX_H0: bounds of config.h0_lo / h0_hi = where the reference's own search stops
                                                    # indexing its 750-row grid (search_time_range -36.4 .. +22.6 s, receiver.py:346-347)
MIN_H0_FD, MAX_H0_FD = -140, 220                    # FT8RX_MIN_H0_FD / _MAX_H0_FD: candidates inside take the frequency-domain fine sync
MAX_F0, GRID_COLS_WIDE, SPEC_BINS_WIDE, MAX_F0_WIDE = 960, 1920, 96000, 1888      # Handle.grid_cols / .spec_bins hold the loaded variant's
MAX_CANDS_WIDE = 2048                               # FT8RX_MAX_CANDS of the wide build: more than any search range has f0 bins


class Config(C.Structure):
    _fields_ = [("sync_score_min", C.c_float), ("max_cands", C.c_int32),
                ("f0_lo", C.c_int32), ("f0_hi", C.c_int32), ("h0_lo", C.c_int32), ("h0_hi", C.c_int32),
                ("bp_nc0_a", C.c_int32), ("bp_iters_a", C.c_int32), ("bp_nc0_b", C.c_int32), ("bp_iters_b", C.c_int32),
                ("osd_single", C.c_int32), ("osd_double", C.c_int32), ("llr_sd_min", C.c_float),
                ("osd_triple", C.c_int32), ("osd_max_hd", C.c_int32)]


RECORD_DTYPE = np.dtype([("msg_lo", "<u8"), ("msg_hi", "<u8"), ("score", "<f4"), ("grid_sd", "<f4"), ("fine_sd", "<f4"),
                         ("f0_idx", "<i2"), ("h0_idx", "<i2"), ("ttweak", "i1"), ("ftweak", "i1"), ("snr_grid", "i1"),
                         ("snr_fine", "i1"), ("status", "u1"), ("ipass", "u1"), ("ap", "u1"), ("method", "u1"),
                         ("n_its", "<i2"), ("nsync", "u1"), ("osd_hd", "u1"), ("pad2", "<u4")])
EVENT_DTYPE = np.dtype([("msg_lo", "<u8"), ("msg_hi", "<u8"), ("cand", "<u2"), ("ipass", "u1"), ("slot", "u1"),
                        ("seq", "<u2"), ("valid", "<u2")])
MESSAGE_DTYPE = np.dtype([("f", "S16", (3,)), ("cand", "<i2"), ("f0_idx", "<i2"), ("h0_idx", "<i2"), ("snr", "i1"), ("ttweak", "i1"),
                          ("ftweak", "i1"), ("ipass", "u1"), ("ap", "u1"), ("method", "u1"), ("fine", "u1"), ("pad", "u1", (3,))])
SUBSIG_DTYPE = np.dtype([("fHz", "<f8"), ("tsec", "<f8"), ("tones", "u1", (79,)), ("pad", "u1")])
assert SUBSIG_DTYPE.itemsize == 96
# packed results (include/ft8rx.h: ft8rx_packed_header / ft8rx_packed_frame): header | frame table | kept records | used events
PACKED_MAGIC = 0x50385446
PACKED_HEADER_DTYPE = np.dtype([("magic", "<u4"), ("n_frames", "<i4"), ("n_records", "<i4"), ("n_events", "<i4"), ("bytes", "<u8"),
                                ("max_cands", "<i4"), ("overflow", "<i4")])
PACKED_FRAME_DTYPE = np.dtype([("rec_off", "<i4"), ("ev_off", "<i4"), ("n_cand", "<u2"), ("n_rec", "<u2"), ("n_ev", "<i4")])
assert PACKED_HEADER_DTYPE.itemsize == 32 and PACKED_FRAME_DTYPE.itemsize == 16
assert RECORD_DTYPE.itemsize == 48 and EVENT_DTYPE.itemsize == 24 and MESSAGE_DTYPE.itemsize == 64

ST_ACTIVE, ST_DECODED, ST_STOP_GRID_SD, ST_STOP_COSTAS, ST_STOP_FINE_SD, ST_EXHAUSTED = range(6)
M_GOOD91, M_LDPC_A, M_LDPC_B, M_OSD, M_LDPC_B_OSD = range(5)

_libs = {}
_reject_log = [None]                  # set_reject_log's current path: applied to builds loaded later as well


class Ft8rxError(RuntimeError):
    pass


def _build_one(path, extra=(), ilp_flags=None, verbose=False):
    """Two objects (different scheduling strategies), one shared library at `path`.  If the ILP unit does not compile WITH its
    scheduling flag -- a hidden LLVM option a ROCm update may rename -- it is retried without (results are bit-identical either way,
    the flag is worth ~3 % on two kernels); objects and stray compiler processes are cleaned up whatever happens."""
    flags = [f for f in HIPCC_FLAGS if f != "-shared"]
    ilp = list(ILP_FLAGS if ilp_flags is None else ilp_flags)
    objs = [path[:-3] + ".main.o", path[:-3] + ".ilp.o"]
    os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
    cmds = [["hipcc"] + flags + list(extra) + ["-c", "-o", objs[0], SRC],
            ["hipcc"] + flags + ilp + list(extra) + ["-c", "-o", objs[1], SRC_ILP]]
    link = ["hipcc", "--offload-arch=gfx950", "-fPIC", "-shared", "-o", path] + objs
    procs = []
    try:
        if verbose:
            for c in cmds + [link]:
                print(" ".join(c))
        procs = [subprocess.Popen(c) for c in cmds]
        rcs = [p.wait() for p in procs]
        if rcs[1] != 0 and ilp:
            import warnings
            warnings.warn(f"hipcc rejected the ILP unit with {' '.join(ilp)}; building it with the default scheduler", RuntimeWarning)
            cmds[1] = [a for a in cmds[1] if a not in ilp]
            rcs[1] = subprocess.call(cmds[1])
        for c, rc in zip(cmds, rcs):
            if rc != 0:
                raise subprocess.CalledProcessError(rc, c)
        subprocess.check_call(link)
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
                p.wait()
        for o in objs:
            if os.path.exists(o):
                os.remove(o)
    return path


def build(force=False, verbose=False):
    """Compile the HIP library for gfx950 in-tree (hipcc cross-compiles without a GPU): libft8rx.so and libft8rx_wide.so."""
    deps = [os.path.join(os.path.dirname(HERE), "include", "ft8rx.h")]
    for d, _, files in os.walk(os.path.join(HERE, "csrc")):            # ft8rx.hip + ft8_dev.h, ft8_tables.h, kernels/*.hpp, host_messages.hpp
        deps += [os.path.join(d, f) for f in files if f.endswith((".hip", ".h", ".hpp"))]
    newest = max(os.path.getmtime(d) for d in deps)
    todo = [(path, extra) for path, extra in ((LIB_PATH, []), (LIB_PATH_WIDE, ["-DFT8RX_WIDE"]))
            if force or not os.path.exists(path) or os.path.getmtime(path) < newest]
    if len(todo) == 2:                                       # both variants side by side (four compiler processes)
        import threading
        errs = []

        def run(path, extra):
            try:
                _build_one(path, extra, verbose=verbose)
            except Exception as e:
                errs.append(e)
        ts = [threading.Thread(target=run, args=t) for t in todo]
        for t in ts:
            t.start()
        for t in ts:
            t.join()
        if errs:
            raise errs[0]
    else:
        for path, extra in todo:
            _build_one(path, extra, verbose=verbose)
    return LIB_PATH


def build_variant(path, extra=(), ilp_flags=None):
    """An alternative build of the library (same two-unit recipe) at `path`, e.g. build_variant("build/ab/x.so", ["-DFINE_TIMING"]) for
    A/B timing through FT8RX_LIB (tools/ab_full.sh) -- never the product path."""
    try:
        return _build_one(path, extra, ilp_flags)
    except subprocess.CalledProcessError as e:
        raise Ft8rxError(f"build_variant({path}) failed: {e}")


def lib(wide=False):
    """The loaded library; wide=True -> the build with the wide layouts (Handle picks it when cfg.f0_hi > 960 or cfg.max_cands > 256).
    The host-only entry points (tone encoder, hash tables, defaults) are the same code in both and are taken from the default one; the
    message layer's capacity follows FT8RX_MAX_CANDS, so record arrays wider than 256 candidates are packaged by the wide build's copy."""
    wide = bool(wide)
    if wide not in _libs:
        path = LIB_PATH_WIDE if wide else LIB_PATH
        if not os.path.exists(path):
            raise Ft8rxError(f"{path} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                             "(pyft8_amd has no CPU fallback)")
        # PyTorch-ROCm wheels bundle their own libamdhip64.so.7; libft8rx.so links the system one (/opt/rocm, same soname).  The
        # copy loaded first serves both: this library runs on either, torch only on its own -- so if torch is already imported,
        # let it load and initialise its runtime before ours is pulled in.
        import sys
        if "torch" in sys.modules:
            try:
                sys.modules["torch"].cuda.is_available()
            except Exception:
                pass
        L = C.CDLL(path)
        L.ft8rx_last_error.restype = C.c_char_p
        L.ft8rx_last_error.argtypes = [C.c_void_p]
        L.ft8rx_create.argtypes = [C.POINTER(Config), C.c_int, C.c_int, C.POINTER(C.c_void_p)]
        L.ft8rx_destroy.argtypes = [C.c_void_p]
        L.ft8rx_destroy.restype = None
        L.ft8rx_staging_audio.restype = C.c_void_p
        L.ft8rx_staging_audio.argtypes = [C.c_void_p]
        gc, sb, mf = C.c_int32(), C.c_int32(), C.c_int32()
        L.ft8rx_build_info(C.byref(gc), C.byref(sb), C.byref(mf))
        want = (GRID_COLS_WIDE, SPEC_BINS_WIDE, MAX_F0_WIDE) if wide else (GRID_COLS, SPEC_BINS, MAX_F0)
        if (gc.value, sb.value, mf.value) != want:
            raise Ft8rxError(f"{path} was built with layouts {(gc.value, sb.value, mf.value)}, expected {want}")
        mc, ec = C.c_int32(), C.c_int32()
        L.ft8rx_build_limits(C.byref(mc), C.byref(ec))
        if (mc.value, ec.value) != ((MAX_CANDS_WIDE if wide else MAX_CANDS), EVENT_CAP):
            raise Ft8rxError(f"{path} was built with capacities {(mc.value, ec.value)} (FT8RX_MAX_CANDS, FT8RX_EVENT_CAP)")
        _libs[wide] = L
        if _reject_log[0]:                                   # a reject log set before this build was loaded applies to it too
            L.ft8rx_set_reject_log.argtypes = [C.c_char_p]
            L.ft8rx_set_reject_log(_reject_log[0].encode())
    return _libs[wide]


def default_config(**kw):
    c = Config()
    lib().ft8rx_default_config(C.byref(c))
    for k, v in kw.items():
        setattr(c, k, v)
    return c


def device_pci_bus_id(device):
    """PCI address ("0000:c1:00.0") of HIP device `device` (ft8rx_device_pci_bus_id), or None."""
    buf = C.create_string_buffer(64)
    L = lib()
    L.ft8rx_device_pci_bus_id.argtypes = [C.c_int, C.c_char_p, C.c_int]
    return buf.value.decode() if L.ft8rx_device_pci_bus_id(int(device), buf, 64) == 0 else None


def fft_plans():
    ps = [(C.c_int32 * 8)() for _ in range(4)]
    lib().ft8rx_get_fft_plans(*ps)
    names = ["plan1920", "plan3200", "plan300", "plan320"]
    return {n: [x for x in p if x] for n, p in zip(names, ps)}


def _ptr(a, t):
    return a.ctypes.data_as(C.POINTER(t))


class Handle:
    """One HIP device + stream + preallocated workspaces for up to max_frames frames."""

    def __init__(self, cfg=None, device=0, max_frames=1):
        self.cfg = cfg or default_config()
        self.wide = self.cfg.f0_hi > MAX_F0 or self.cfg.max_cands > MAX_CANDS      # beyond 3000 Hz / 256 candidates: the wide build (include/ft8rx.h)
        L = self._L = lib(self.wide)
        self.grid_cols, self.spec_bins = (GRID_COLS_WIDE, SPEC_BINS_WIDE) if self.wide else (GRID_COLS, SPEC_BINS)
        self.max_frames = int(max_frames)
        self.device = int(device)
        self._h = C.c_void_p()
        rc = L.ft8rx_create(C.byref(self.cfg), self.device, self.max_frames, C.byref(self._h))
        if rc != 0:
            raise Ft8rxError(f"ft8rx_create failed ({rc}): {L.ft8rx_last_error(None).decode()}")

    def close(self):
        if getattr(self, "_h", None) is not None and self._h.value:
            self._L.ft8rx_destroy(self._h)
            self._h = C.c_void_p()
            self._packed_keep = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _chk(self, rc, what):
        if rc != 0:
            raise Ft8rxError(f"{what} failed ({rc}): {self._L.ft8rx_last_error(self._h).decode()}")

    # ---- whole path
    def decode_batch(self, audio):
        audio = np.ascontiguousarray(audio, np.int16)
        if audio.ndim == 1:
            audio = audio[None]
        B = audio.shape[0]
        if audio.ndim != 2 or audio.shape[1] != NSAMP:
            raise Ft8rxError(f"audio must be int16 [n_frames, {NSAMP}] (15 s at 12 kHz); got shape {audio.shape} -- see receiver.frames_from_ragged")
        if B < 1:
            raise Ft8rxError("empty batch")
        return self._run(audio, B)

    def _alloc_out(self, B):
        # records / counts are written in full by the library; of the event rows only the used entries are (the rest must read as zero)
        mc = self.cfg.max_cands
        return (np.empty((B, mc), RECORD_DTYPE), np.empty(B, np.int32), np.zeros((B, EVENT_CAP), EVENT_DTYPE), np.empty(B, np.int32))

    def _run(self, audio, B):
        rec, cnt, ev, evc = self._alloc_out(B)
        rc = self._L.ft8rx_decode_batch(self._h, _ptr(audio, C.c_int16), B, rec.ctypes.data_as(C.c_void_p), _ptr(cnt, C.c_int32),
                                      ev.ctypes.data_as(C.c_void_p), _ptr(evc, C.c_int32))
        self._chk(rc, "ft8rx_decode_batch")
        return rec, cnt, ev, evc

    def decode_messages(self, audio, max_msgs=None, n_threads=None, table=None, return_flags=False):
        """ft8rx_decode_messages: host audio -> (messages[B, max_msgs] of MESSAGE_DTYPE, counts[B]) in one native call.  An
        overflowed event log or message list is warned about (Ft8rxTruncationWarning) and reported per frame with return_flags."""
        audio = np.ascontiguousarray(audio, np.int16)
        if audio.ndim == 1:
            audio = audio[None]
        B = audio.shape[0]
        if audio.shape[1] != NSAMP or B > self.max_frames:
            raise Ft8rxError(f"audio must be [n<={self.max_frames}, {NSAMP}] int16, got {audio.shape}")
        max_msgs = int(max_msgs or max(1, self.cfg.max_cands))
        out = np.zeros((B, max_msgs), MESSAGE_DTYPE)
        oc = np.zeros(B, np.int32)
        flags = np.zeros(B, np.int32)
        L = self._L
        L.ft8rx_decode_messages.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
        self._chk(L.ft8rx_decode_messages(self._h, audio.ctypes.data, int(B), out.ctypes.data, max_msgs, oc.ctypes.data,
                                          int(n_threads or min(32, os.cpu_count() or 1)), table._t if table is not None else None,
                                          flags.ctypes.data), "ft8rx_decode_messages")
        _warn_truncation(flags, "decode_messages")
        return (out, oc, flags) if return_flags else (out, oc)

    def enqueue(self, d_audio_ptr, B):
        self._chk(self._L.ft8rx_enqueue_batch(self._h, C.c_void_p(d_audio_ptr), int(B)), "ft8rx_enqueue_batch")

    def enqueue_host(self, audio):
        """Asynchronous decode of host audio (int16 [B, 180000], ideally from pinned_audio()): ft8rx_enqueue_batch_host.  The array
        must stay alive and unchanged until the batch has been fetched."""
        if audio.dtype != np.int16 or audio.ndim != 2 or audio.shape[1] != NSAMP or not audio.flags["C_CONTIGUOUS"]:
            raise Ft8rxError(f"enqueue_host: audio must be a C-contiguous int16 [n_frames, {NSAMP}] array")
        L = self._L
        L.ft8rx_enqueue_batch_host.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
        self._chk(L.ft8rx_enqueue_batch_host(self._h, audio.ctypes.data_as(C.c_void_p), int(audio.shape[0])), "ft8rx_enqueue_batch_host")

    def sync(self):
        self._chk(self._L.ft8rx_sync(self._h), "ft8rx_sync")

    def fetch(self, B, out=None):
        """ft8rx_fetch_results: the oldest unfetched batch's results, copied into fresh arrays -- or into `out` = a (rec, cnt, ev, evc)
        tuple an earlier call returned (a loop that fetches every few milliseconds then allocates nothing; event entries beyond
        the counts are stale in a reused set)."""
        rec, cnt, ev, evc = out if out is not None else self._alloc_out(B)
        if out is not None and (rec.shape != (B, self.cfg.max_cands) or ev.shape != (B, EVENT_CAP) or rec.dtype != RECORD_DTYPE or ev.dtype != EVENT_DTYPE
                                or cnt.shape != (B,) or evc.shape != (B,) or cnt.dtype != np.int32 or evc.dtype != np.int32
                                or not all(a.flags.c_contiguous and a.flags.writeable for a in (rec, cnt, ev, evc))):
            raise Ft8rxError("fetch: `out` is not a result set of this handle and batch size")
        rc = self._L.ft8rx_fetch_results(self._h, int(B), rec.ctypes.data_as(C.c_void_p), _ptr(cnt, C.c_int32),
                                       ev.ctypes.data_as(C.c_void_p), _ptr(evc, C.c_int32))
        self._chk(rc, "ft8rx_fetch_results")
        return rec, cnt, ev, evc

    def fetch_view(self, B):
        """Like fetch, without the copy: numpy views of the handle's page-locked result buffers (ft8rx_fetch_results_view).
        Valid until two more batches have been enqueued."""
        B = int(B)
        p = [C.c_void_p() for _ in range(4)]
        L = self._L
        L.ft8rx_fetch_results_view.argtypes = [C.c_void_p, C.c_int] + [C.POINTER(C.c_void_p)] * 4
        self._chk(L.ft8rx_fetch_results_view(self._h, B, *[C.byref(x) for x in p]), "ft8rx_fetch_results_view")
        mc = self.cfg.max_cands

        def view(ptr, nbytes, dtype, shape):
            return np.frombuffer((C.c_char * nbytes).from_address(ptr.value), dtype=dtype).reshape(shape)
        return (view(p[0], B * mc * RECORD_DTYPE.itemsize, RECORD_DTYPE, (B, mc)), view(p[1], B * 4, np.int32, (B,)),
                view(p[2], B * EVENT_CAP * EVENT_DTYPE.itemsize, EVENT_DTYPE, (B, EVENT_CAP)), view(p[3], B * 4, np.int32, (B,)))

    def results_to_device(self, B, d_rec, d_cnt, d_ev, d_evc):
        """Latest batch's results -> caller-owned device buffers (raw device pointers; ft8rx_results_to_device)."""
        L = self._L
        L.ft8rx_results_to_device.argtypes = [C.c_void_p, C.c_int] + [C.c_void_p] * 4
        self._chk(L.ft8rx_results_to_device(self._h, int(B), C.c_void_p(d_rec), C.c_void_p(d_cnt), C.c_void_p(d_ev), C.c_void_p(d_evc)),
                  "ft8rx_results_to_device")

    def set_packed_output(self, buf0, buf1, cap_bytes, keep=None):
        """ft8rx_set_packed_output: raw pointers (device memory, or page-locked host memory from pinned_bytes()) of the two buffers the
        following batches write their packed results into (one per result slot); None, None turns it off.
        keep: the objects that own the two buffers (torch tensors, page-locked arrays).  The handle holds on to them -- and to the events
        given to packed_fence -- until the packed output is reset or the handle is closed, so the pack kernels can never write into
        memory whose Python owner has already been collected."""
        L = self._L
        L.ft8rx_set_packed_output.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64]
        self._chk(L.ft8rx_set_packed_output(self._h, C.c_void_p(buf0 or None), C.c_void_p(buf1 or None), C.c_uint64(int(cap_bytes))),
                  "ft8rx_set_packed_output")
        # (the call above has waited for every batch in flight: the previous owners may go now)
        self._packed_keep = {"buffers": keep, "fence": [None, None]} if (buf0 or buf1) else None

    def packed_fence(self, which, hip_event, keep=None):
        """ft8rx_packed_output_fence: the next batch that packs into buffer `which` waits (on the device) for this HIP event, e.g.
        torch.cuda.Event(...).cuda_event recorded behind an asynchronous send of the buffer.  keep: the object that owns the event."""
        L = self._L
        L.ft8rx_packed_output_fence.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
        self._chk(L.ft8rx_packed_output_fence(self._h, int(which), C.c_void_p(hip_event or None)), "ft8rx_packed_output_fence")
        if getattr(self, "_packed_keep", None) is not None:
            self._packed_keep["fence"][int(which)] = keep

    def d2h_async(self, dst_ptr, src_ptr, nbytes):
        """ft8rx_d2h_async: device -> page-locked host copy on the handle's result-copy stream; -> ticket (d2h_done / d2h_event)."""
        L = self._L
        L.ft8rx_d2h_async.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.POINTER(C.c_int32)]
        t = C.c_int32()
        self._chk(L.ft8rx_d2h_async(self._h, C.c_void_p(dst_ptr), C.c_void_p(src_ptr), C.c_uint64(int(nbytes)), C.byref(t)), "ft8rx_d2h_async")
        return int(t.value)

    def d2h_done(self, ticket):
        L = self._L
        L.ft8rx_d2h_query.argtypes = [C.c_void_p, C.c_int32]
        r = L.ft8rx_d2h_query(self._h, int(ticket))
        if r < 0:
            raise Ft8rxError(f"ft8rx_d2h_query failed ({r}): {L.ft8rx_last_error(self._h).decode()}")
        return r == 1

    def d2h_event(self, ticket):
        L = self._L
        L.ft8rx_d2h_event.argtypes = [C.c_void_p, C.c_int32]
        L.ft8rx_d2h_event.restype = C.c_void_p
        return L.ft8rx_d2h_event(self._h, int(ticket))

    def packed_results(self):
        """ft8rx_packed_results: (which of the two packed buffers, its header as a dict) for the batch the last fetch returned."""
        which = C.c_int32()
        hdr = np.zeros(1, PACKED_HEADER_DTYPE)
        L = self._L
        L.ft8rx_packed_results.argtypes = [C.c_void_p, C.POINTER(C.c_int32), C.c_void_p]
        self._chk(L.ft8rx_packed_results(self._h, C.byref(which), hdr.ctypes.data), "ft8rx_packed_results")
        return int(which.value), {k: int(hdr[0][k]) for k in PACKED_HEADER_DTYPE.names}

    def pinned_bytes(self, nbytes):
        """uint8 array of page-locked host memory (ft8rx_alloc_host), released with the array."""
        L = self._L
        L.ft8rx_alloc_host.restype = C.c_void_p
        L.ft8rx_alloc_host.argtypes = [C.c_void_p, C.c_uint64]
        L.ft8rx_free_host.argtypes = [C.c_void_p, C.c_void_p]
        p = L.ft8rx_alloc_host(self._h, int(nbytes))
        if not p:
            raise Ft8rxError(f"ft8rx_alloc_host failed: {L.ft8rx_last_error(self._h).decode()}")
        buf = (C.c_uint8 * int(nbytes)).from_address(p)
        arr = np.frombuffer(buf, dtype=np.uint8)
        weakref.finalize(buf, L.ft8rx_free_host, None, C.c_void_p(p))
        return arr

    def set_streams(self, n):
        self._chk(self._L.ft8rx_set_streams(self._h, int(n)), "ft8rx_set_streams")

    def set_subbatch(self, frames):
        """Frames per kernel chain inside a stream's share of a batch (ft8rx_set_subbatch; default 256, 0 = the whole share at once)."""
        self._chk(self._L.ft8rx_set_subbatch(self._h, int(frames)), "ft8rx_set_subbatch")

    def set_ladder_mode(self, mode):
        """0 (default) = fine-stage BP in ladder order, three launches (throughput); 1 = one launch for the five AP variants (latency)."""
        self._chk(self._L.ft8rx_set_ladder_mode(self._h, int(mode)), "ft8rx_set_ladder_mode")

    def set_search_mask(self, mask):
        """mask[n_frames, f0_hi - f0_lo] (non-zero = search this column, any score > 0) for the following batches, or None = the
        configured search again (ft8rx_set_search_mask: the subtraction experiment's local re-search)."""
        if mask is None:
            self._chk(self._L.ft8rx_set_search_mask(self._h, None, 0), "ft8rx_set_search_mask")
            return
        m = np.ascontiguousarray(mask, np.uint8)
        nf0 = self.cfg.f0_hi - self.cfg.f0_lo
        if m.ndim != 2 or m.shape[1] != nf0:
            raise Ft8rxError(f"set_search_mask: mask must be [n_frames, {nf0}]")
        self._chk(self._L.ft8rx_set_search_mask(self._h, m.ctypes.data_as(C.c_void_p), m.shape[0]), "ft8rx_set_search_mask")

    def set_profiling(self, on):
        self._L.ft8rx_set_profiling(self._h, int(bool(on)))

    def stage_times(self):
        n = C.c_int()
        names = (C.c_char_p * 24)()
        ms = (C.c_float * 24)()
        self._L.ft8rx_get_stage_times(self._h, C.byref(n), names, ms)
        return {names[i].decode(): ms[i] for i in range(n.value)}

    # ---- stage entry points
    def spectrogram(self, audio):
        audio = np.ascontiguousarray(audio, np.int16)
        if audio.ndim == 1:
            audio = audio[None]
        B = audio.shape[0]
        g = np.empty((B, GRID_ROWS, self.grid_cols), np.float32)
        self._chk(self._L.ft8rx_spectrogram(self._h, _ptr(audio, C.c_int16), B, _ptr(g, C.c_float)), "ft8rx_spectrogram")
        return g

    def hop_spectrum(self, window3840):
        w = np.ascontiguousarray(window3840, np.int16)
        if w.shape != (3840,):
            raise Ft8rxError(f"hop_spectrum needs the last 3840 samples, got shape {w.shape}")
        row = np.empty(self.grid_cols, np.float32)
        self._chk(self._L.ft8rx_hop_spectrum(self._h, _ptr(w, C.c_int16), _ptr(row, C.c_float)), "ft8rx_hop_spectrum")
        return row

    def _grid(self, grid):
        grid = np.ascontiguousarray(grid, np.float32)
        if grid.ndim == 2:
            grid = grid[None]
        if grid.shape[1:] != (GRID_ROWS, self.grid_cols):
            raise Ft8rxError(f"grid must be [n][{GRID_ROWS}][{self.grid_cols}] for this handle, got {grid.shape}")
        return grid

    def sync_search(self, grid):
        grid = self._grid(grid)
        B, mc = grid.shape[0], self.cfg.max_cands
        f0 = np.zeros((B, mc), np.int32); h0 = np.zeros((B, mc), np.int32); sc = np.zeros((B, mc), np.float32); cnt = np.zeros(B, np.int32)
        self._chk(self._L.ft8rx_sync_search(self._h, _ptr(grid, C.c_float), B, _ptr(f0, C.c_int32), _ptr(h0, C.c_int32),
                                          _ptr(sc, C.c_float), _ptr(cnt, C.c_int32)), "ft8rx_sync_search")
        return f0, h0, sc, cnt

    def sync_scores(self, grid, f0_lo, f0_hi):
        """ft8rx_sync_scores: per frame and f0 in [f0_lo, f0_hi) the best Costas score (first strict maximum over h0, from 0) and its h0."""
        grid = self._grid(grid)
        B, n = grid.shape[0], int(f0_hi) - int(f0_lo)
        sc = np.zeros((B, max(n, 1)), np.float32); h0 = np.zeros((B, max(n, 1)), np.int32)
        L = self._L
        L.ft8rx_sync_scores.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
        self._chk(L.ft8rx_sync_scores(self._h, grid.ctypes.data, B, int(f0_lo), int(f0_hi), sc.ctypes.data, h0.ctypes.data), "ft8rx_sync_scores")
        return sc, h0

    def llr_grid(self, grid, frame, f0, h0):
        grid = self._grid(grid)
        frame, f0, h0 = (np.ascontiguousarray(x, np.int32) for x in (frame, f0, h0))
        n = len(f0)
        llr = np.zeros((n, 174), np.float32); sd = np.zeros(n, np.float32); snr = np.zeros(n, np.int32)
        self._chk(self._L.ft8rx_llr_grid(self._h, _ptr(grid, C.c_float), grid.shape[0], n, _ptr(frame, C.c_int32), _ptr(f0, C.c_int32),
                                       _ptr(h0, C.c_int32), _ptr(llr, C.c_float), _ptr(sd, C.c_float), _ptr(snr, C.c_int32)), "ft8rx_llr_grid")
        return llr, sd, snr

    def cycle_spectrum(self, audio):
        audio = np.ascontiguousarray(audio, np.int16)
        if audio.ndim == 1:
            audio = audio[None]
        B = audio.shape[0]
        s = np.empty((B, self.spec_bins), np.complex64)
        self._chk(self._L.ft8rx_cycle_spectrum(self._h, _ptr(audio, C.c_int16), B, s.ctypes.data_as(C.POINTER(C.c_float))), "ft8rx_cycle_spectrum")
        return s

    def fine(self, spec, frame, f0, h0, want_sgrid=False):
        spec = np.ascontiguousarray(spec, np.complex64)
        if spec.ndim == 1:
            spec = spec[None]
        if spec.shape[1] != self.spec_bins:
            raise Ft8rxError(f"spectrum must be [n][{self.spec_bins}] for this handle, got {spec.shape}")
        frame, f0, h0 = (np.ascontiguousarray(x, np.int32) for x in (frame, f0, h0))
        n = len(f0)
        ret, tt, ft, ns, snr = (np.zeros(n, np.int32) for _ in range(5))
        llr = np.zeros((n, 174), np.float32); sd = np.zeros(n, np.float32)
        sg = np.zeros((n, 79, 8), np.float32) if want_sgrid else None
        self._chk(self._L.ft8rx_fine(self._h, spec.ctypes.data_as(C.POINTER(C.c_float)), spec.shape[0], n, _ptr(frame, C.c_int32),
                                   _ptr(f0, C.c_int32), _ptr(h0, C.c_int32), _ptr(ret, C.c_int32), _ptr(tt, C.c_int32), _ptr(ft, C.c_int32),
                                   _ptr(ns, C.c_int32), _ptr(llr, C.c_float), _ptr(sd, C.c_float), _ptr(snr, C.c_int32),
                                   _ptr(sg, C.c_float) if want_sgrid else None), "ft8rx_fine")
        return dict(ret=ret, ttweak=tt, ftweak=ft, nsync=ns, llr=llr, sd=sd, snr=snr, sgrid=sg)

    def ldpc(self, llr, max_ncheck0, max_iters):
        llr = np.ascontiguousarray(llr, np.float32).reshape(-1, 174)
        n = len(llr)
        ok, nits, has = (np.zeros(n, np.int32) for _ in range(3))
        lo, hi = np.zeros(n, np.uint64), np.zeros(n, np.uint64)
        out = np.zeros((n, 174), np.float32)
        self._chk(self._L.ft8rx_ldpc(self._h, _ptr(llr, C.c_float), n, int(max_ncheck0), int(max_iters), _ptr(ok, C.c_int32),
                                   _ptr(lo, C.c_uint64), _ptr(hi, C.c_uint64), _ptr(nits, C.c_int32), _ptr(has, C.c_int32),
                                   _ptr(out, C.c_float)), "ft8rx_ldpc")
        return ok, lo, hi, nits, has, out

    def osd(self, llr, singleflips=30, doubleflips=2, tripleflips=0, max_hd=0, want_hd=False):
        llr = np.ascontiguousarray(llr, np.float32).reshape(-1, 174)
        n = len(llr)
        ok, trial, hd = np.zeros(n, np.int32), np.zeros(n, np.int32), np.zeros(n, np.int32)
        lo, hi = np.zeros(n, np.uint64), np.zeros(n, np.uint64)
        self._chk(self._L.ft8rx_osd_ext(self._h, _ptr(llr, C.c_float), n, int(singleflips), int(doubleflips), int(tripleflips), int(max_hd),
                                      _ptr(ok, C.c_int32), _ptr(lo, C.c_uint64), _ptr(hi, C.c_uint64), _ptr(trial, C.c_int32),
                                      _ptr(hd, C.c_int32)), "ft8rx_osd_ext")
        return (ok, lo, hi, trial, hd) if want_hd else (ok, lo, hi, trial)

    def crc_valid(self, cw91):
        cw91 = np.ascontiguousarray(cw91, np.float32).reshape(-1, 91)
        n = len(cw91)
        res = np.zeros(n, np.int32); lo, hi = np.zeros(n, np.uint64), np.zeros(n, np.uint64)
        self._chk(self._L.ft8rx_crc_valid(self._h, _ptr(cw91, C.c_float), n, _ptr(res, C.c_int32), _ptr(lo, C.c_uint64), _ptr(hi, C.c_uint64)), "ft8rx_crc_valid")
        return res, lo, hi

    def valid77(self, bits):
        lo = np.array([b & (2 ** 64 - 1) for b in bits], np.uint64)
        hi = np.array([b >> 64 for b in bits], np.uint64)
        out = np.zeros(len(lo), np.int32)
        self._chk(self._L.ft8rx_valid77(self._h, _ptr(lo, C.c_uint64), _ptr(hi, C.c_uint64), len(lo), _ptr(out, C.c_int32)), "ft8rx_valid77")
        return out

    def subtract(self, d_audio_ptr, n_frames, signals, return_float=False, refine=False, return_origins=False):
        """Subtract decoded signals from device-resident int16 audio in place (ft8rx_subtract).
        signals: per frame a list of (tones79, fHz, tsec), subtracted in list order.  -> float32 residual if return_float.
        refine: 0 = as given (the reference's subtract_signal), 1 = re-estimate each origin first with full-rate scans, 2 = the same on
        a decimated baseband copy (fast; what Receiver's multi-pass decode uses)."""
        B = int(n_frames)
        if isinstance(signals, tuple):          # (array [B, max_sigs] of SUBSIG_DTYPE, counts [B]) -- the fast path
            arr, cnt = signals
            arr = np.ascontiguousarray(arr, SUBSIG_DTYPE)
            cnt = np.ascontiguousarray(cnt, np.int32)
            ms = arr.shape[1]
        else:
            ms = max(1, max((len(s) for s in signals), default=1))
            arr = np.zeros((B, ms), SUBSIG_DTYPE)
            cnt = np.zeros(B, np.int32)
            for f, lst in enumerate(signals):
                cnt[f] = len(lst)
                for i, (tones, fHz, tsec) in enumerate(lst):
                    arr[f, i]["tones"] = np.asarray(tones, np.uint8)
                    arr[f, i]["fHz"], arr[f, i]["tsec"] = fHz, tsec
        out = np.empty((B, NSAMP), np.float32) if return_float else None
        L = self._L
        L.ft8rx_subtract.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p]
        self._chk(L.ft8rx_subtract(self._h, C.c_void_p(d_audio_ptr), B, arr.ctypes.data_as(C.c_void_p), _ptr(cnt, C.c_int32), ms,
                                   int(refine), out.ctypes.data_as(C.c_void_p) if return_float else None), "ft8rx_subtract")
        if return_origins:                 # (fHz, tsec) per signal after refinement
            return out, [[(float(arr[f, i]["fHz"]), float(arr[f, i]["tsec"])) for i in range(cnt[f])] for f in range(B)]
        return out

    def pinned_audio(self, n_frames):
        """int16 [n_frames, 180000] array in page-locked host memory (ft8rx_alloc_host): fill it and pass it to decode_batch for
        overlapped DMA.  The memory is released when the array (and every view of it) is garbage collected."""
        L = self._L
        L.ft8rx_alloc_host.restype = C.c_void_p
        L.ft8rx_alloc_host.argtypes = [C.c_void_p, C.c_uint64]
        L.ft8rx_free_host.argtypes = [C.c_void_p, C.c_void_p]
        nbytes = int(n_frames) * NSAMP * 2
        p = L.ft8rx_alloc_host(self._h, nbytes)
        if not p:
            raise Ft8rxError(f"ft8rx_alloc_host failed: {L.ft8rx_last_error(self._h).decode()}")
        buf = (C.c_int16 * (int(n_frames) * NSAMP)).from_address(p)
        arr = np.frombuffer(buf, dtype=np.int16).reshape(int(n_frames), NSAMP)
        weakref.finalize(buf, L.ft8rx_free_host, None, C.c_void_p(p))
        return arr

    def staging_ptr(self):
        return int(self._L.ft8rx_staging_audio(self._h))

    def download_audio(self, d_ptr, n_frames):
        out = np.empty((n_frames, NSAMP), np.int16)
        self._chk(self._L.ft8rx_copy_to_host(self._h, out.ctypes.data_as(C.c_void_p), C.c_void_p(d_ptr), C.c_uint64(out.nbytes)), "ft8rx_copy_to_host")
        return out

    def synth_frames(self, d_audio_ptr, start, count, n_signals=50, snr_range=(-10.0, 10.0), seed=0x4654385F53594E54, noise=True,
                     return_table=False):
        """Fill the device buffer at d_audio_ptr ([count][180000] int16) with synthetic frames; returns the truth list (and the
        signal table handed to the kernel if return_table).  noise=False leaves the Philox noise out (parity tests)."""
        from . import synth
        recs, truth = synth.device_signal_table(start, count, n_signals, snr_range)
        assert recs.dtype.itemsize == synth.SIGNAL_DTYPE.itemsize
        q = np.ascontiguousarray(synth.pulse_cumsum(), np.float64)
        self._chk(self._L.ft8rx_synth_frames_ex(self._h, C.c_uint64(seed), int(start), int(count), int(n_signals),
                                              recs.ctypes.data_as(C.c_void_p), int(recs.dtype.itemsize), _ptr(q, C.c_double),
                                              C.c_void_p(d_audio_ptr), int(not noise)), "ft8rx_synth_frames")
        return (truth, recs) if return_table else truth

    def math_probe(self, which, x):
        if which == 2:
            x = np.ascontiguousarray(x, np.complex64)
            y = np.empty_like(x)
            self._chk(self._L.ft8rx_math_probe(self._h, 2, x.ctypes.data_as(C.POINTER(C.c_float)), len(x), y.ctypes.data_as(C.POINTER(C.c_float))), "ft8rx_math_probe")
            return y
        x = np.ascontiguousarray(x, np.float32)
        y = np.empty_like(x)
        self._chk(self._L.ft8rx_math_probe(self._h, int(which), _ptr(x, C.c_float), x.size, _ptr(y, C.c_float)), "ft8rx_math_probe")
        return y


PKG_MSG_TRUNCATED, PKG_EVENTS_TRUNCATED = 1, 2


class Ft8rxTruncationWarning(RuntimeWarning):
    """A frame's event log (FT8RX_EVENT_CAP) or message list overflowed: see include/ft8rx.h FT8RX_PKG_*."""


class CallHashTable:
    """Persistent native call-hash table (ft8rx_hashes_*; reference databases.py:8-26) for package_batch(table=...)."""

    def __init__(self):
        L = lib()
        L.ft8rx_hashes_create.restype = C.c_void_p
        L.ft8rx_hashes_destroy.argtypes = [C.c_void_p]
        L.ft8rx_hashes_destroy.restype = None
        L.ft8rx_hashes_clear.argtypes = [C.c_void_p]
        L.ft8rx_hashes_add.argtypes = [C.c_void_p, C.c_char_p]
        L.ft8rx_hashes_size.argtypes = [C.c_void_p]
        self._t = C.c_void_p(L.ft8rx_hashes_create())
        if not self._t.value:
            raise Ft8rxError("ft8rx_hashes_create failed")

    def add(self, call):
        lib().ft8rx_hashes_add(self._t, call.encode())

    def clear(self):
        lib().ft8rx_hashes_clear(self._t)

    def __len__(self):
        return int(lib().ft8rx_hashes_size(self._t))

    def __del__(self):
        try:
            if self._t.value:
                lib().ft8rx_hashes_destroy(self._t)
                self._t = C.c_void_p()
        except Exception:
            pass


def set_reject_log(path):
    """Turn the reference's rejected_callsigns.txt side effect (decoders.py:114-115) on (path) or off (None) -- in every loaded build
    of the library (the packager inside Handle.decode_messages is the wide build's own copy on a wide handle)."""
    _reject_log[0] = path
    lib()
    for L in _libs.values():
        L.ft8rx_set_reject_log.argtypes = [C.c_char_p]
        L.ft8rx_set_reject_log(path.encode() if path else None)


def _warn_truncation(flags, who):
    if flags.any():
        import warnings
        nev, nmsg = int((flags & PKG_EVENTS_TRUNCATED != 0).sum()), int((flags & PKG_MSG_TRUNCATED != 0).sum())
        warnings.warn(f"{who}: event log overflowed in {nev} frame(s) (> {EVENT_CAP} CRC-passing words: `<...>` strings may differ), "
                      f"message list truncated in {nmsg} frame(s)", Ft8rxTruncationWarning, stacklevel=3)


def package_batch(rec, cnt, ev, evc, max_msgs=None, n_threads=None, table=None, return_flags=False, out=None):
    """Native host message layer (ft8rx_package_batch): records/events of B frames -> (messages[B, max_msgs], counts[B]).
    Pure host code: works without a GPU.  max_msgs defaults to the record capacity (so the list cannot be truncated); table = a
    CallHashTable shared by the frames in order (streaming) instead of a fresh table per frame.  An overflowed event log or
    message list raises Ft8rxTruncationWarning (warnings module) and is reported per frame in the flags (return_flags=True).
    out = the (messages, counts, flags) arrays of an earlier call with return_flags=True: written in place instead of fresh arrays
    (message slots beyond the counts are stale then)."""
    rec = np.ascontiguousarray(rec)
    ev = np.ascontiguousarray(ev)
    cnt = np.ascontiguousarray(cnt, np.int32)
    evc = np.ascontiguousarray(evc, np.int32)
    B, mc = rec.shape
    if ev.shape != (B, EVENT_CAP) or rec.dtype != RECORD_DTYPE or ev.dtype != EVENT_DTYPE:
        raise Ft8rxError("package_batch: records/events are not the arrays returned by decode_batch/fetch")
    if max_msgs is None:
        max_msgs = max(mc, 1)
    if out is not None:
        out, oc, flags = out
        if (out.shape != (B, max_msgs) or out.dtype != MESSAGE_DTYPE or oc.shape != (B,) or oc.dtype != np.int32 or flags.shape != (B,)
                or flags.dtype != np.int32 or not all(a.flags.c_contiguous and a.flags.writeable for a in (out, oc, flags))):
            raise Ft8rxError("package_batch: `out` does not fit this batch")
    else:
        out = np.zeros((B, max_msgs), MESSAGE_DTYPE)
        oc = np.zeros(B, np.int32)
        flags = np.zeros(B, np.int32)
    if n_threads is None:
        n_threads = min(32, os.cpu_count() or 1)
    L = lib(mc > MAX_CANDS)
    L.ft8rx_package_batch.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p,
                                      C.c_int, C.c_void_p, C.c_void_p]
    rc = L.ft8rx_package_batch(rec.ctypes.data, cnt.ctypes.data, ev.ctypes.data, evc.ctypes.data, int(B), int(mc), out.ctypes.data,
                               int(max_msgs), oc.ctypes.data, int(n_threads), table._t if table is not None else None, flags.ctypes.data)
    if rc != 0:
        raise Ft8rxError(f"ft8rx_package_batch failed ({rc})")
    _warn_truncation(flags, "package_batch")
    return (out, oc, flags) if return_flags else (out, oc)


def packed_capacity(n_frames, max_cands=MAX_CANDS, per_frame=None):
    """Bytes a packed result buffer needs for n_frames frames: the worst case (every candidate kept, full event logs) by default, or
    per_frame bytes of records + events per frame (config 1 frames use ~4 KB; an overflow is flagged, never silent)."""
    worst = RECORD_DTYPE.itemsize * int(max_cands) + EVENT_DTYPE.itemsize * EVENT_CAP
    body = worst if per_frame is None else min(worst, int(per_frame))
    return PACKED_HEADER_DTYPE.itemsize + int(n_frames) * (PACKED_FRAME_DTYPE.itemsize + body)


def pack_results(rec, cnt, ev, evc):
    """Host twin of the k_pack_* kernels: dense result arrays (as returned by fetch / decode_batch) -> one uint8 array in the packed
    layout of include/ft8rx.h.  Used where results are host arrays already (the gloo gather of tests, tools)."""
    rec = np.ascontiguousarray(rec)
    B, mc = rec.shape
    cnt = np.clip(np.asarray(cnt, np.int64), 0, mc)
    evc = np.asarray(evc, np.int64)
    nev = np.clip(evc, 0, EVENT_CAP)
    keep = np.zeros((B, mc), bool)
    for f in range(B):
        n = int(cnt[f])
        k = rec[f, :n]["status"] == ST_DECODED
        c = np.asarray(ev[f, :int(nev[f])]["cand"], np.int64)
        k[c[c < n]] = True
        if np.isnan(rec[f, :n]["grid_sd"]).any() or np.isnan(rec[f, :n]["fine_sd"]).any():
            k[:] = True
        keep[f, :n] = k
    nrec = keep.sum(1)
    table = np.zeros(B, PACKED_FRAME_DTYPE)
    table["rec_off"] = np.concatenate([[0], np.cumsum(nrec)[:-1]]) if B else []
    table["ev_off"] = np.concatenate([[0], np.cumsum(nev)[:-1]]) if B else []
    table["n_cand"], table["n_rec"], table["n_ev"] = cnt, nrec, np.maximum(evc, 0)
    recs = rec[keep].copy()                                   # row-major boolean take: frame order, candidate order inside a frame
    recs["pad2"] = np.nonzero(keep)[1]
    evs = np.concatenate([ev[f, :int(nev[f])] for f in range(B)]) if B else np.zeros(0, EVENT_DTYPE)
    hdr = np.zeros(1, PACKED_HEADER_DTYPE)
    hdr["magic"], hdr["n_frames"], hdr["n_records"], hdr["n_events"], hdr["max_cands"] = PACKED_MAGIC, B, len(recs), len(evs), mc
    hdr["bytes"] = hdr.nbytes + table.nbytes + recs.nbytes + evs.nbytes
    return np.concatenate([hdr.view(np.uint8), table.view(np.uint8), recs.view(np.uint8).reshape(-1), evs.view(np.uint8).reshape(-1)])


class Packed:
    """Read-only view of a packed result buffer (any object with the buffer protocol; nothing is copied): .header, .frames (the
    frame table), .records, .events, frame(f) -> (records, events) of one frame."""

    def __init__(self, buf):
        b = np.frombuffer(buf, np.uint8) if not isinstance(buf, np.ndarray) else buf.view(np.uint8).reshape(-1)
        if b.size < PACKED_HEADER_DTYPE.itemsize:
            raise Ft8rxError("packed results: buffer shorter than a header")
        self.header = b[:32].view(PACKED_HEADER_DTYPE)[0]
        h = self.header
        if int(h["magic"]) != PACKED_MAGIC:
            raise Ft8rxError("packed results: bad magic")
        if int(h["overflow"]):
            raise Ft8rxError(f"packed results: {int(h['bytes'])} bytes did not fit the buffer handed to set_packed_output")
        nf, nr, ne = int(h["n_frames"]), int(h["n_records"]), int(h["n_events"])
        o1 = 32 + 16 * nf
        o2 = o1 + RECORD_DTYPE.itemsize * nr
        o3 = o2 + EVENT_DTYPE.itemsize * ne
        if o3 != int(h["bytes"]) or o3 > b.size:
            raise Ft8rxError("packed results: inconsistent sizes")
        self.nbytes = o3
        self.buf = b[:o3]
        self.n_frames, self.max_cands = nf, int(h["max_cands"])
        self.frames = b[32:o1].view(PACKED_FRAME_DTYPE)
        self.records = b[o1:o2].view(RECORD_DTYPE)
        self.events = b[o2:o3].view(EVENT_DTYPE)

    def frame(self, f):
        t = self.frames[f]
        return (self.records[int(t["rec_off"]):int(t["rec_off"]) + int(t["n_rec"])],
                self.events[int(t["ev_off"]):int(t["ev_off"]) + min(int(t["n_ev"]), EVENT_CAP)])

    def expand(self):
        """-> dense (records[B, max_cands], counts[B], events[B, EVENT_CAP], event_counts[B]) with the kept records at their candidate
        positions; every other record is zero (tests, tools)."""
        B, mc = self.n_frames, self.max_cands
        rec = np.zeros((B, mc), RECORD_DTYPE)
        ev = np.zeros((B, EVENT_CAP), EVENT_DTYPE)
        for f in range(B):
            r, e = self.frame(f)
            idx = r["pad2"].astype(np.int64)
            rr = r.copy()
            rr["pad2"] = 0
            rec[f, idx] = rr
            ev[f, :len(e)] = e
        return rec, self.frames["n_cand"].astype(np.int32), ev, self.frames["n_ev"].astype(np.int32)


def package_packed(buf, frame_lo=0, n_frames=None, max_msgs=None, n_threads=None, table=None, return_flags=False):
    """ft8rx_package_packed (host only): the messages of frames [frame_lo, frame_lo + n_frames) of a packed result buffer, as
    package_batch renders them from the dense arrays."""
    pk = buf if isinstance(buf, Packed) else Packed(buf)
    n = pk.n_frames - frame_lo if n_frames is None else int(n_frames)
    if max_msgs is None:
        max_msgs = max(pk.max_cands, 1)
    out = np.zeros((max(n, 0), max_msgs), MESSAGE_DTYPE)
    oc = np.zeros(max(n, 0), np.int32)
    flags = np.zeros(max(n, 0), np.int32)
    if n > 0:
        L = lib(pk.max_cands > MAX_CANDS)
        L.ft8rx_package_packed.argtypes = [C.c_void_p, C.c_uint64, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
        rc = L.ft8rx_package_packed(pk.buf.ctypes.data, C.c_uint64(pk.nbytes), int(frame_lo), n, out.ctypes.data, int(max_msgs), oc.ctypes.data,
                                    int(n_threads or min(32, os.cpu_count() or 1)), table._t if table is not None else None, flags.ctypes.data)
        if rc != 0:
            raise Ft8rxError(f"ft8rx_package_packed failed ({rc})")
        _warn_truncation(flags, "package_packed")
    return (out, oc, flags) if return_flags else (out, oc)


def merge_messages(out, out_counts, add, add_counts, pass_tag, drop_osd=False):
    """ft8rx_merge_messages: append (in place) the messages of a later pass that are new for their frame; -> (fresh, fresh_counts),
    the appended messages without the pass tag."""
    if out.dtype != MESSAGE_DTYPE or add.dtype != MESSAGE_DTYPE or not out.flags.c_contiguous or out.shape[0] != add.shape[0]:
        raise Ft8rxError("merge_messages: message arrays as returned by package_batch expected")
    add = np.ascontiguousarray(add)
    add_counts = np.ascontiguousarray(add_counts, np.int32)
    if out_counts.dtype != np.int32 or not out_counts.flags.c_contiguous:
        raise Ft8rxError("merge_messages: out_counts must be a contiguous int32 array (it is updated in place)")
    fresh = np.zeros_like(add)
    fc = np.zeros(add.shape[0], np.int32)
    L = lib()
    L.ft8rx_merge_messages.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int,
                                       C.c_void_p, C.c_void_p]
    rc = L.ft8rx_merge_messages(out.ctypes.data, out_counts.ctypes.data, int(out.shape[1]), add.ctypes.data, add_counts.ctypes.data,
                                int(add.shape[1]), int(out.shape[0]), int(pass_tag), int(bool(drop_osd)), fresh.ctypes.data, fc.ctypes.data)
    if rc != 0:
        raise Ft8rxError(f"ft8rx_merge_messages failed ({rc})")
    return fresh, fc


def subtraction_list(msgs, mcnt, rec, min_snr):
    """ft8rx_subtraction_list (host only): the signals a subtraction sweep removes -- every message with snr > min_snr, in emit order,
    as (tones, fHz, tsec).  -> (signals[B, max] of SUBSIG_DTYPE, counts[B])."""
    import math
    msgs = np.ascontiguousarray(msgs)
    rec = np.ascontiguousarray(rec)
    mcnt = np.ascontiguousarray(mcnt, np.int32)
    if msgs.dtype != MESSAGE_DTYPE or rec.dtype != RECORD_DTYPE or msgs.shape[0] != rec.shape[0]:
        raise Ft8rxError("subtraction_list: message / record arrays as returned by package_batch / decode_batch expected")
    B = msgs.shape[0]
    cap = max(1, int(mcnt.max()) if B else 1)
    arr = np.zeros((B, cap), SUBSIG_DTYPE)
    cnt = np.zeros(B, np.int32)
    L = lib()
    L.ft8rx_subtraction_list.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p]
    most = L.ft8rx_subtraction_list(msgs.ctypes.data, mcnt.ctypes.data, int(msgs.shape[1]), rec.ctypes.data, int(rec.shape[1]), int(B),
                                    int(math.floor(min_snr)), arr.ctypes.data, int(cap), cnt.ctypes.data)
    if most < 0:
        raise Ft8rxError(f"ft8rx_subtraction_list failed ({most})")
    return np.ascontiguousarray(arr[:, :max(1, most)]), cnt


_default = {}


def encode_tones(msg_lo, msg_hi):
    """77-bit words (as returned in records) -> uint8 [n, 79] tone sequences (ft8rx_encode_tones, host only)."""
    lo = np.ascontiguousarray(msg_lo, np.uint64).ravel()
    hi = np.ascontiguousarray(msg_hi, np.uint64).ravel()
    out = np.zeros((len(lo), 79), np.uint8)
    L = lib()
    L.ft8rx_encode_tones.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
    if L.ft8rx_encode_tones(lo.ctypes.data, hi.ctypes.data, len(lo), out.ctypes.data) != 0:
        raise Ft8rxError("ft8rx_encode_tones failed")
    return out


def default_handle(max_frames=1):
    """Process-wide handle used by the function-style API in decoders.py."""
    h = _default.get("h")
    if h is None or h.max_frames < max_frames:
        if h is not None:
            h.close()
        h = Handle(max_frames=max_frames)
        _default["h"] = h
    return h
