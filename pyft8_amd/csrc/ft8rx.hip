// ft8rx.hip -- MI355X (gfx950) FT8 receive hot path: HIP kernels + the C ABI of include/ft8rx.h.
//
// Pipeline for a batch of B independent 15-s frames (stream-ordered, no host
// round trips; frames are the batch dimension, candidates the second one):
//   k_spectrogram  (hop, frame)         Hann * 3840-pt real FFT (1920-pt complex Stockham in LDS) -> dB grid
//   k_sync         (16-f0 tile, frame)  Costas correlation over all time offsets from an LDS tile
//   k_topk         (frame)              threshold + stable top-K (bitonic sort in LDS)
//   k_grid_llr     (candidate)          payload gather -> max-log LLRs -> sigma normalisation
//   k_bp           (candidate, AP)      one wavefront: GOOD91 + flooding BP, ballot parity, LDS exchange
//   k_select0      (candidate)          first success in ladder order
//   k_cyc_a/bc     (tile, frame)        192000-pt real FFT as 300x320 four-step; row pass fused with the real split
//   k_fine         (candidate)          9x(slice/taper/3200-pt IFFT) + 32-pt DFT scoring, Costas gate, LLRs
//   k_bp           (candidate, AP)      GOOD91 + BP(90,20) with saved outputs
//   k_select1, k_osd (candidate, slot) one wavefront: rank sort, register-resident GF(2) Gauss-Jordan
//                                       with ballot pivoting, lane-per-trial CRC-14 + validity, k_select2
// A batch is cut into chunks whose chains run on separate HIP streams -- free-running: chunk i of batch k+1 follows chunk i of batch k
// on stream i, no per-batch fork / join; results land in one of two result slots and are copied to page-locked host buffers by a copy
// stream while the next batch computes; the used part of the event log is packed by k_ev_scan / k_ev_compact straight into
// page-locked host memory (launch_batch / ft8rx_fetch_results).
// k_sub_* / k_subd_* subtract decoded signals (ft8rx_subtract; refine 0 = the reference experiment's subtract_signal,
// 3 = its refine_time_origin first: k_cyc_a_f32 + k_refine3; 1 / 2 = the build's own origin re-estimation).
// Reference line citations are to PyFT8/receiver.py and PyFT8/decoders.py (see include/ft8rx.h).
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <new>
#include <string>
#include <vector>
#include "../../include/ft8rx.h"
#include "ft8_dev.h"

#define MAXC FT8RX_MAX_CANDS
#define WL_MULT(i) ((i) == WL_BP0 ? 5 : (i) == WL_OSDNAN ? 10 : 1)      /* entries per candidate slot of work list i */
#define NF0MAX (FT8RX_MAX_F0 > 1024 ? 2048 : 1024)     /* per-frame stride of the per-f0 sync results; k_topk sorts this many keys */

// The kernels live in one file per stage; this file is the C ABI, the handle and the launch chains.
#include "kernels/common.hpp"
#include "kernels/spectrogram.hpp"
#include "kernels/sync_search.hpp"
#include "kernels/llr.hpp"
#include "kernels/bp.hpp"
#include "kernels/cycle_spectrum.hpp"
#include "kernels/fine_sync.hpp"
#include "kernels/osd.hpp"
#include "kernels/synth.hpp"
#include "kernels/subtract.hpp"
#include "kernels/probes.hpp"
#include "host_messages.hpp"
#include "ilp_launch.hpp"       // k_fine, k_spectrogram and k_hop_spectrum are launched from the second translation unit (ft8rx_ilp.hip)

// ====================================================================================== host side
#define EV_EAGER_BYTES ((size_t)1 << 20)      /* event logs up to this size travel whole with the records (launch_batch) */
#define HIPCHK(h, x) do { hipError_t _e = (x); if (_e != hipSuccess) { set_err(h, "%s failed: %s (%s:%d)", #x, hipGetErrorString(_e), __FILE__, __LINE__); return -2; } } while (0)

// OSD trial list in the reference's order (decoders.py:248-272): order 0; single flips i < S; the restricted double flips
// (i, j), i < S, j < min(i, D), i-major; then the build's order-3 extension (i, j, k), k < j < i < T, i-major.  One packed entry
// per trial: i | j << 8 | k << 16, OSD_NONE for "no flip".
static std::vector<uint32_t> osd_trial_table(int S, int D, int T) {
    std::vector<uint32_t> t;
    const uint32_t N = OSD_NONE;
    t.push_back(N | (N << 8) | (N << 16));
    for (int i = 0; i < S; i++) t.push_back((uint32_t)i | (N << 8) | (N << 16));
    for (int i = 0; i < S; i++) for (int j = 0; j < D && j < i; j++) t.push_back((uint32_t)i | ((uint32_t)j << 8) | (N << 16));
    for (int i = 0; i < T; i++) for (int j = 0; j < i; j++) for (int k = 0; k < j; k++) t.push_back((uint32_t)i | ((uint32_t)j << 8) | ((uint32_t)k << 16));
    return t;
}
static int osd_nflip(int S, int T) { return S > T ? S : T; }

struct ft8rx_hashes { hostmsg::Hashes H; };      // persistent call-hash table (databases.py:8): ft8rx_hashes_* in include/ft8rx.h

// k_sync holds every grid row its h0 window can touch in LDS (244 B per row): windows of at most SYNC_WIN time offsets per launch.  A
// wider search_time_range runs as consecutive windows in ascending h0, each launch keeping the earlier windows' maximum unless it
// finds a strictly larger one -- the first strict maximum of the whole range (receiver.py:350-351), as one launch gives it.
#define SYNC_WIN 352
static size_t sync_lds_bytes(const ft8rx_config& c) {
    const int nh0 = c.h0_hi - c.h0_lo;
    const size_t nrows = (size_t)((nh0 < SYNC_WIN ? nh0 : SYNC_WIN) + 24);
    return nrows * 16 * sizeof(double) + (nrows * 29 + 512) * sizeof(float);
}
static std::string g_create_err;

struct ft8rx_handle {
    ft8rx_config cfg;
    int device, max_frames;
    hipStream_t stream;
    int n_streams;                       // chunks of a batch run their kernel chains on separate streams
    int ladder_mode;                     // fine-stage BP launches: 0 = ladder order (three launches), 1 = one launch (ft8rx_set_ladder_mode)
    int sub_frames;                      // frames per kernel chain inside a chunk (ft8rx_set_subbatch; 0 = the whole chunk in one chain)
    hipStream_t sub[8];
    hipEvent_t ev_fork, ev_join[8];
    hipStream_t copy_s;              // host-to-device chunk copies of ft8rx_decode_batch, in order, never queued behind kernels
    hipEvent_t ev_chunk[16];
    Tables T;
    std::vector<void*> allocs;
    struct Chunk { void* p; size_t cap, used; };
    std::vector<Chunk> arena;        // scratch of the stage entry points (struct Scratch)
    int16_t* d_audio;            // staging for host-pointer entry points
    int16_t* d_audio2;           // second staging buffer (allocated on first use): ft8rx_enqueue_batch_host double-buffers the H2D copies
    hipStream_t h2d_s;           // host-to-device copies of the pipelined host entry: never queued behind kernels or result copies
    float* d_grid;
    float* d_best_score; int32_t* d_best_h0;
    ft8rx_record* d_rec; int32_t* d_ncand;
    float* d_llr0; float* d_saved;
    Att *d_att0, *d_attG, *d_attB, *d_attO;
    cpx *d_A, *d_spec;
    ft8rx_event* d_ev; int32_t* d_evcount;
    const uint32_t* d_trials; int n_trials;      // OSD trial list of this configuration
    int32_t* d_work[WL_N];                       // ladder work lists (kernels/common.hpp: WorkList), [B][MAXC] candidate ids each
    int32_t* d_wcount;                           // [16 chunks][WL_N] list lengths, zeroed at the head of every chunk's chain
    uint8_t* d_colmask; bool use_mask;           // [B][NF0MAX] search mask of the local re-search (ft8rx_set_search_mask), allocated on first use
    // Result slots.  A batch writes its records/events into slot k % 2 (slot 0 = d_rec/d_ncand/d_ev/d_evcount above) and, when its
    // kernels are done, the copy stream moves them into page-locked host buffers while the next batch computes into the other
    // slot; ft8rx_fetch_results hands out the oldest unfetched batch.  At most two batches' results are retained.
    ft8rx_record* s_rec[2]; int32_t* s_ncand[2]; ft8rx_event* s_ev[2]; int32_t* s_evcount[2];
    ft8rx_record* h_rec[2]; int32_t* h_cnt[2]; ft8rx_event* h_ev[2]; int32_t* h_evc[2];
    hipEvent_t ev_comp[2], ev_done[2];
    int slot_enq, slot_fetch, inflight, last_slot, slot_B[2];
    // free-running chunk streams (launch_batch): the chunk chains of consecutive batches follow each other on their own streams
    // without a per-batch fork / join; need_barrier = something else has used the shared workspaces (or the partition changed) since
    hipEvent_t ev_cdone[2][8];
    bool free_running, need_barrier; int part_B, part_n;
    // Large batches (event log > EV_EAGER_BYTES): k_ev_scan / k_ev_compact pack the used entries of the log and write them straight
    // into page-locked host memory (h_evpacked; d_evpacked = the same buffer's device address); the fetch spreads them over h_ev's
    // [frame][cap] rows (fetch_events)
    bool slot_evpending[2];
    int32_t* d_evoffs[2]; ft8rx_event* h_evpacked[2]; ft8rx_event* d_evpacked[2];
    // packed results (ft8rx_set_packed_output): the caller's buffers (as device addresses), one per result slot, the scratch of the
    // pack kernels and a page-locked mirror of each slot's header
    unsigned char* pk_buf[2]; uint64_t pk_cap; uint64_t* d_pkneed; int32_t* d_pknrec;
    ft8rx_packed_header* h_pkhdr[2]; ft8rx_packed_header* d_pkhdr[2]; bool slot_packed[2]; int fetched_slot;
    hipEvent_t pk_fence[2];                      // ft8rx_packed_output_fence: the consumer's "done reading this buffer" event
    hipEvent_t d2h_ev[32]; uint32_t d2h_next;    // ft8rx_d2h_async: tickets (a ring of events on the result-copy stream)
    // signal subtraction (extension, allocated on first use): float32 working copy, per-chunk partial sums, GFSK tables
    float* d_wf; double2* d_part; double* d_pulse; double* d_pc; ft8rx_subsig* d_sigs; int32_t* d_sigcnt; int sig_cap;
    float2 *d_zdec, *d_model, *d_adec; SubdCtx* d_subctx;      // decimated-baseband refinement (refine = 2), allocated on first use
    double* d_ones;                                             // refine = 3: an all-ones taper table
    std::string err;
    bool profiling;
    std::vector<hipEvent_t> pev;
    std::vector<const char*> pnames;
    int n_stage;
    float stage_ms[24];
};

static void set_err(ft8rx_handle* h, const char* fmt, ...) {
    char buf[512];
    va_list ap; va_start(ap, fmt); vsnprintf(buf, sizeof(buf), fmt, ap); va_end(ap);
    if (h) h->err = buf; else g_create_err = buf;
}

// Streams are a PROCESS-WIDE pool per device, created once and never destroyed.  The HIP runtime maps every stream onto one of four
// hardware queues when it is created, and commands of streams that share a queue execute in submission order.  The streams of the
// first handle of a process get a queue each (main, copy, H2D, second chunk stream); a handle created after that one had been
// destroyed used to get another mapping, with both chunk streams of a batch on ONE queue -- no overlap between the two halves of a
// batch: 82.4 k -> 70.9 k frames/s on the 8192-frame shard (tools/r05_alloc_probe.py, profiles/r05_notes.md).  Handles of one process
// share the pool (more ordering between them, never less: every wait refers to an event recorded earlier in host order).
#include <mutex>
struct StreamPool { hipStream_t main = nullptr, copy = nullptr, h2d = nullptr, sub[8] = {}; };
static std::mutex g_pool_mu;
static StreamPool g_pool[64];
static hipStream_t pool_stream(int device, hipStream_t StreamPool::* which, bool blocking) {
    std::lock_guard<std::mutex> lk(g_pool_mu);
    StreamPool& P = g_pool[device & 63];
    if (!(P.*which)) {
        hipStream_t s = nullptr;
        if ((blocking ? hipStreamCreate(&s) : hipStreamCreateWithFlags(&s, hipStreamNonBlocking)) != hipSuccess) return nullptr;
        P.*which = s;
    }
    return P.*which;
}
static hipStream_t pool_sub(int device, int i) {
    std::lock_guard<std::mutex> lk(g_pool_mu);
    StreamPool& P = g_pool[device & 63];
    if (!P.sub[i] && hipStreamCreateWithFlags(&P.sub[i], hipStreamNonBlocking) != hipSuccess) P.sub[i] = nullptr;
    return P.sub[i];
}

template <typename T> static int dalloc(ft8rx_handle* h, T** p, size_t n) {
    void* q = nullptr;
    hipError_t e = hipMalloc(&q, n * sizeof(T));
    if (e != hipSuccess) { set_err(h, "hipMalloc(%zu bytes) failed: %s", n * sizeof(T), hipGetErrorString(e)); return -2; }
    h->allocs.push_back(q); *p = (T*)q; return 0;
}

// The handle's own device audio buffer ([max_frames][180000] int16: staging of the host-pointer entry points, ft8rx_staging_audio) is
// allocated when something first asks for it: a caller that keeps its audio resident in HBM (ft8rx_enqueue_batch) never pays the
// 360 KB per frame.
static int need_staging(ft8rx_handle* h) {
    if (h->d_audio) return 0;
    return dalloc(h, &h->d_audio, (size_t)h->max_frames * FT8RX_NSAMP);
}

static void host_twiddle(int n, int count, std::vector<cpx>& w) {
    w.resize(count);
    for (int t = 0; t < count; t++) {
        double ang = (2.0 * M_PI * (double)t) / (double)n;
        w[t] = make_float2((float)cos(ang), (float)(-sin(ang)));
    }
}
template <typename T> static int upload(ft8rx_handle* h, const T** dst, const std::vector<T>& v) {
    T* p; if (dalloc(h, &p, v.size())) return -2;
    if (hipMemcpy(p, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice) != hipSuccess) { set_err(h, "table upload failed"); return -2; }
    *dst = p; return 0;
}

// Device scratch of the stage entry points: bump allocation out of chunks the handle keeps (h->arena), so that after the first call
// of a given size ft8rx_ldpc / ft8rx_osd / ... allocate nothing (the reference-style function API in pyft8_amd/decoders.py calls
// them once per vector).  A Scratch object resets the arena when it goes out of scope; chunks are released by ft8rx_destroy.
struct Scratch {
    ft8rx_handle* h;
    ~Scratch() { for (auto& c : h->arena) c.used = 0; }
    void* raw(size_t bytes) {
        bytes = (bytes + 255) & ~(size_t)255;
        if (!bytes) bytes = 256;
        for (auto& c : h->arena) if (c.cap - c.used >= bytes) { void* q = (char*)c.p + c.used; c.used += bytes; return q; }
        // miss: a new chunk, grown geometrically (twice the largest so far) so that a sequence of calls with growing n settles after
        // O(log n) allocations; chunks no call is using any more are released first instead of piling up until ft8rx_destroy
        size_t largest = 0;
        for (auto& c : h->arena) if (c.cap > largest) largest = c.cap;
        size_t cap = bytes > 2 * largest ? bytes : 2 * largest;
        if (cap < ((size_t)1 << 20)) cap = (size_t)1 << 20;
        for (size_t i = 0; i < h->arena.size();) {
            if (h->arena[i].used == 0) { hipFree(h->arena[i].p); h->arena.erase(h->arena.begin() + i); } else i++;
        }
        void* q = nullptr;
        if (hipMalloc(&q, cap) != hipSuccess) { if (cap == bytes || hipMalloc(&q, bytes) != hipSuccess) return nullptr; cap = bytes; }
        h->arena.push_back({q, cap, bytes});
        return q;
    }
    template <typename T> T* get(size_t n) { return (T*)raw(n * sizeof(T)); }
    template <typename T> T* put(const T* src, size_t n) { T* q = get<T>(n); if (q && hipMemcpy(q, src, n * sizeof(T), hipMemcpyHostToDevice) != hipSuccess) return nullptr; return q; }
};

// Everything that is not a pipelined batch (stage entry points, the synchronous host entry, subtraction, profiling passes) uses the
// handle's shared workspaces from the main stream; while the chunk streams run free (launch_batch) their work is not ordered
// against the main stream, so such calls first wait until the batches in flight are complete.
static int quiesce(ft8rx_handle* h, bool next_is_batch = false) {
    // the result copy of the last batch waits for every chunk stream (free-running batches) and still reads the slot's device
    // buffers on the copy stream after a plain one -- either way nothing of a batch is pending once the copy stream is idle.
    // A plain BATCH after a plain batch need not wait on the host at all: its kernels follow the previous ones on the main stream (the
    // sub-streams were joined into it), it writes the OTHER result slot, and that slot's previous copy is ordered by ev_done[slot] --
    // so small pipelined batches (enqueue k + 1 while k's results travel) keep their overlap.
    if (h->free_running || (h->last_slot >= 0 && !next_is_batch)) HIPCHK(h, hipStreamSynchronize(h->copy_s));
    h->free_running = false;
    h->need_barrier = true;
    return 0;
}
#define ENTER(h) do { HIPCHK(h, hipSetDevice((h)->device)); { const int _q = quiesce(h); if (_q) return _q; } } while (0)

extern "C" {

int ft8rx_default_config(ft8rx_config* c) {
    if (!c) return -1;
    c->sync_score_min = 85.0f; c->max_cands = 200; c->f0_lo = 32; c->f0_hi = 960; c->h0_lo = -37; c->h0_hi = 87;
    c->bp_nc0_a = 35; c->bp_iters_a = 5; c->bp_nc0_b = 90; c->bp_iters_b = 20; c->osd_single = 30; c->osd_double = 2;
    c->llr_sd_min = 5.0f;
    c->osd_triple = 0; c->osd_max_hd = 0;
    return 0;
}

int ft8rx_device_count(void) { int n = 0; if (hipGetDeviceCount(&n) != hipSuccess) return 0; return n; }
int ft8rx_device_pci_bus_id(int device, char* buf, int len) {
    if (!buf || len < 16) return -1;
    return hipDeviceGetPCIBusId(buf, len, device) == hipSuccess ? 0 : -2;
}
int ft8rx_build_info(int32_t* grid_cols, int32_t* spec_bins, int32_t* max_f0) {
    static_assert(NF0MAX >= FT8RX_GRID_COLS && NF0MAX >= FT8RX_MAX_F0, "per-f0 scratch stride");
    if (grid_cols) *grid_cols = FT8RX_GRID_COLS;
    if (spec_bins) *spec_bins = FT8RX_SPEC_BINS;
    if (max_f0) *max_f0 = FT8RX_MAX_F0;
    return 0;
}
int ft8rx_build_limits(int32_t* max_cands, int32_t* event_cap) {
    static_assert(MAXC % 256 == 0 && MAXC <= NF0MAX, "candidate stride: whole 256-thread blocks, never more than k_topk has keys");
    if (max_cands) *max_cands = FT8RX_MAX_CANDS;
    if (event_cap) *event_cap = FT8RX_EVENT_CAP;
    return 0;
}

int ft8rx_get_fft_plans(int32_t* p1920, int32_t* p3200, int32_t* p300, int32_t* p320) {
    const int32_t a[8] = {8, 4, 4, 5, 3, 0, 0, 0}, b[8] = {8, 4, 4, 5, 5, 0, 0, 0}, c[8] = {5, 5, 4, 3, 0, 0, 0, 0}, d[8] = {8, 8, 5, 0, 0, 0, 0, 0};
    if (p1920) memcpy(p1920, a, sizeof(a));
    if (p3200) memcpy(p3200, b, sizeof(b));
    if (p300) memcpy(p300, c, sizeof(c));
    if (p320) memcpy(p320, d, sizeof(d));
    return 0;
}

const char* ft8rx_last_error(ft8rx_handle* h) { return h ? h->err.c_str() : g_create_err.c_str(); }

void ft8rx_destroy(ft8rx_handle* h) {
    if (!h) return;
    hipSetDevice(h->device);
    // the streams belong to the process-wide pool and stay: wait for everything this handle has put on them, then free its memory
    if (h->stream) hipStreamSynchronize(h->stream);
    if (h->copy_s) hipStreamSynchronize(h->copy_s);
    if (h->h2d_s) hipStreamSynchronize(h->h2d_s);
    for (int i = 0; i < 8; i++) { if (h->sub[i]) hipStreamSynchronize(h->sub[i]); if (h->ev_join[i]) hipEventDestroy(h->ev_join[i]); }
    for (void* p : h->allocs) hipFree(p);
    for (auto& c : h->arena) hipFree(c.p);
    for (auto e : h->pev) hipEventDestroy(e);
    for (int k = 0; k < 2; k++) for (int i = 0; i < 8; i++) if (h->ev_cdone[k][i]) hipEventDestroy(h->ev_cdone[k][i]);
    for (int i = 0; i < 32; i++) if (h->d2h_ev[i]) hipEventDestroy(h->d2h_ev[i]);
    for (int i = 0; i < 16; i++) if (h->ev_chunk[i]) hipEventDestroy(h->ev_chunk[i]);
    if (h->ev_fork) hipEventDestroy(h->ev_fork);
    for (int k = 0; k < 2; k++) {
        if (h->ev_comp[k]) hipEventDestroy(h->ev_comp[k]);
        if (h->ev_done[k]) hipEventDestroy(h->ev_done[k]);
        if (h->h_rec[k]) hipHostFree(h->h_rec[k]);
        if (h->h_cnt[k]) hipHostFree(h->h_cnt[k]);
        if (h->h_ev[k]) hipHostFree(h->h_ev[k]);
        if (h->h_evc[k]) hipHostFree(h->h_evc[k]);
        if (h->h_evpacked[k]) hipHostFree(h->h_evpacked[k]);
        if (h->h_pkhdr[k]) hipHostFree(h->h_pkhdr[k]);
    }
    delete h;
}

int ft8rx_create(const ft8rx_config* cfg, int device, int max_frames, ft8rx_handle** out) {
    if (!cfg || !out || max_frames < 1) { set_err(nullptr, "ft8rx_create: bad arguments"); return -1; }
    if (cfg->max_cands < 1 || cfg->max_cands > MAXC || cfg->f0_lo < 4 || cfg->f0_hi > FT8RX_MAX_F0 || cfg->f0_lo >= cfg->f0_hi ||
        cfg->h0_hi <= cfg->h0_lo || cfg->h0_lo < FT8RX_MIN_H0 || cfg->h0_hi > FT8RX_MAX_H0 || cfg->bp_nc0_a > cfg->bp_nc0_b || cfg->bp_iters_a > cfg->bp_iters_b ||
        cfg->osd_single < 0 || cfg->osd_single > OSD_MAXFLIP || cfg->osd_double < 0 || cfg->osd_double > OSD_MAXFLIP ||
        cfg->osd_triple < 0 || cfg->osd_triple > 40 || cfg->osd_max_hd < 0 || cfg->osd_max_hd > 174 ||
        osd_trial_table(cfg->osd_single, cfg->osd_double, cfg->osd_triple).size() > OSD_MAXTRIALS) {
        set_err(nullptr, "ft8rx_create: configuration out of the supported range"); return -1; }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1) { set_err(nullptr, "ft8rx_create: no HIP device available (this library has no CPU fallback)"); return -3; }
    if (device < 0 || device >= ndev) { set_err(nullptr, "ft8rx_create: device %d out of range (%d devices)", device, ndev); return -1; }
    ft8rx_handle* h = new ft8rx_handle();
    h->cfg = *cfg; h->device = device; h->max_frames = max_frames; h->stream = nullptr; h->profiling = false; h->n_stage = 0;
    h->n_streams = 2; h->ladder_mode = 0; h->sub_frames = FT8RX_SUBBATCH_DEFAULT; h->ev_fork = nullptr; for (int i = 0; i < 8; i++) { h->sub[i] = nullptr; h->ev_join[i] = nullptr; }
    h->copy_s = nullptr; h->slot_evpending[0] = h->slot_evpending[1] = false; h->h2d_s = nullptr; h->d_audio = nullptr; h->d_audio2 = nullptr; for (int i = 0; i < 16; i++) h->ev_chunk[i] = nullptr;
    for (int k = 0; k < 2; k++) { h->ev_comp[k] = h->ev_done[k] = nullptr; h->h_rec[k] = nullptr; h->h_cnt[k] = nullptr; h->h_ev[k] = nullptr; h->h_evc[k] = nullptr; h->h_evpacked[k] = nullptr; h->d_evpacked[k] = nullptr; h->d_evoffs[k] = nullptr; h->slot_B[k] = 0; }
    h->slot_enq = h->slot_fetch = h->inflight = 0; h->last_slot = -1;
    for (int k = 0; k < 2; k++) for (int i = 0; i < 8; i++) h->ev_cdone[k][i] = nullptr;
    h->free_running = false; h->need_barrier = true; h->part_B = h->part_n = 0;
    h->d_wf = nullptr; h->d_part = nullptr; h->d_pulse = nullptr; h->d_pc = nullptr; h->d_sigs = nullptr; h->d_sigcnt = nullptr; h->sig_cap = 0;
    h->d_colmask = nullptr; h->use_mask = false;
    for (int i = 0; i < 32; i++) h->d2h_ev[i] = nullptr;
    h->d2h_next = 0;
    h->pk_buf[0] = h->pk_buf[1] = nullptr; h->pk_cap = 0; h->d_pkneed = nullptr; h->d_pknrec = nullptr; h->fetched_slot = -1;
    for (int k = 0; k < 2; k++) { h->h_pkhdr[k] = nullptr; h->d_pkhdr[k] = nullptr; h->slot_packed[k] = false; h->pk_fence[k] = nullptr; }
    h->d_zdec = nullptr; h->d_model = nullptr; h->d_adec = nullptr; h->d_subctx = nullptr; h->d_ones = nullptr;
    if (hipSetDevice(device) != hipSuccess || !(h->stream = pool_stream(device, &StreamPool::main, true))) { set_err(nullptr, "ft8rx_create: cannot open device %d", device); delete h; return -2; }
    const size_t B = (size_t)max_frames;
    int rc = 0;
    rc |= dalloc(h, &h->d_grid, B * FT8RX_GRID_ROWS * FT8RX_GRID_COLS);
    rc |= dalloc(h, &h->d_best_score, B * NF0MAX);
    rc |= dalloc(h, &h->d_best_h0, B * NF0MAX);
    rc |= dalloc(h, &h->d_rec, B * MAXC);
    rc |= dalloc(h, &h->d_ncand, B);
    rc |= dalloc(h, &h->d_llr0, B * MAXC * 174);
    rc |= dalloc(h, &h->d_saved, B * MAXC * 5 * 174);
    rc |= dalloc(h, &h->d_att0, B * MAXC * 5);
    rc |= dalloc(h, &h->d_attG, B * MAXC * 2);
    rc |= dalloc(h, &h->d_attB, B * MAXC * 5);
    rc |= dalloc(h, &h->d_attO, B * MAXC * 10);
    // The four-step scratch (768 KB per frame) and the cycle spectrum (393 KB; 768 KB in the wide build) live INSIDE the grid buffer: the
    // dB grid is dead once k_grid_llr has gathered the payloads, and the cycle FFT only starts after that (enqueue_chain places a chunk's
    // A and spectrum at the start of the chunk's own grid region; k_spectrogram rewrites the constant row 0 with every batch).  The
    // stage entry points use one of the three at a time.  -1.16 MB of HBM per frame.
    static_assert((size_t)(96000 + FT8RX_SPEC_BINS) * sizeof(cpx) <= (size_t)FT8RX_GRID_ROWS * FT8RX_GRID_COLS * sizeof(float), "scratch A + cycle spectrum fit a frame's grid");
    if (!rc) { h->d_A = reinterpret_cast<cpx*>(h->d_grid); h->d_spec = h->d_A + B * 96000; }
    for (int i = 0; i < WL_N; i++) rc |= dalloc(h, &h->d_work[i], B * MAXC * WL_MULT(i));      // WL_BP0 / WL_OSDNAN list attempts
    rc |= dalloc(h, &h->d_wcount, (size_t)16 * WL_N);
    rc |= dalloc(h, &h->d_ev, B * FT8RX_EVENT_CAP);
    rc |= dalloc(h, &h->d_evcount, B);
    h->s_rec[0] = h->d_rec; h->s_ncand[0] = h->d_ncand; h->s_ev[0] = h->d_ev; h->s_evcount[0] = h->d_evcount;
    rc |= dalloc(h, &h->s_rec[1], B * MAXC);
    rc |= dalloc(h, &h->s_ncand[1], B);
    rc |= dalloc(h, &h->s_ev[1], B * FT8RX_EVENT_CAP);
    rc |= dalloc(h, &h->s_evcount[1], B);
    const bool ev_compact = B * FT8RX_EVENT_CAP * sizeof(ft8rx_event) > EV_EAGER_BYTES;
    for (int k = 0; k < 2 && ev_compact; k++) rc |= dalloc(h, &h->d_evoffs[k], B + 1);
    for (int k = 0; k < 2 && !rc; k++) {
        bool okh = hipHostMalloc((void**)&h->h_rec[k], sizeof(ft8rx_record) * B * cfg->max_cands, hipHostMallocDefault) == hipSuccess;
        okh = okh && hipHostMalloc((void**)&h->h_cnt[k], sizeof(int32_t) * B, hipHostMallocDefault) == hipSuccess;
        okh = okh && hipHostMalloc((void**)&h->h_ev[k], sizeof(ft8rx_event) * B * FT8RX_EVENT_CAP, hipHostMallocDefault) == hipSuccess;
        okh = okh && hipHostMalloc((void**)&h->h_evc[k], sizeof(int32_t) * B, hipHostMallocDefault) == hipSuccess;
        if (ev_compact) okh = okh && hipHostMalloc((void**)&h->h_evpacked[k], sizeof(ft8rx_event) * B * FT8RX_EVENT_CAP, hipHostMallocDefault) == hipSuccess
                                  && hipHostGetDevicePointer((void**)&h->d_evpacked[k], h->h_evpacked[k], 0) == hipSuccess;
        if (!okh) { set_err(h, "ft8rx_create: page-locked result buffers (%zu frames) could not be allocated", B); rc = -2; }
    }
    if (rc) { g_create_err = h->err; ft8rx_destroy(h); return -2; }
    // ---- tables (double precision on the host, rounded once)
    std::vector<float> win(3840);
    for (int i = 0; i < 3840; i++) win[i] = (float)(0.5 + 0.5 * cos(M_PI * (double)(2 * i + 1 - 3840) / 3839.0));   // np.hanning (receiver.py:236)
    std::vector<double> taper(100);
    { double step = (0.0 - M_PI) / 99.0; for (int i = 0; i < 100; i++) { double y = (i == 99) ? 0.0 : (double)i * step + M_PI; taper[i] = 0.5 * (1.0 + cos(y)); } }   // receiver.py:183-184
    std::vector<cpx> w;
    rc |= upload(h, &h->T.win, win);
    rc |= upload(h, &h->T.taper, taper);
    host_twiddle(1920, 1920, w);   rc |= upload(h, &h->T.W1920, w);
    host_twiddle(3840, FT8RX_GRID_COLS, w);    rc |= upload(h, &h->T.WR3840, w);
    host_twiddle(3200, 3200, w);   rc |= upload(h, &h->T.W3200, w);
    host_twiddle(96000, 96000, w); rc |= upload(h, &h->T.W96000, w);
    host_twiddle(300, 300, w);     rc |= upload(h, &h->T.W300, w);
    host_twiddle(320, 320, w);     rc |= upload(h, &h->T.W320, w);
    host_twiddle(192000, FT8RX_SPEC_BINS, w); rc |= upload(h, &h->T.WR192k, w);
    host_twiddle(32, 32, w);       rc |= upload(h, &h->T.W32, w);
    {   // tables of the frequency-domain fine score (kernels/fine_sync.hpp: fine_fscore; oracle/ft8_oracle.c: make_fscore_tables -- same formulas)
        std::vector<float> k32(1800);
        std::vector<cpx> cs(156), g1000(1000);
        for (int m = -900; m < 900; m++) {
            const int r = ((m % 100) + 100) % 100;
            k32[m + 900] = (r == 0) ? (m == 0 ? 32.0f : 0.0f) : (float)(sin(M_PI * (double)r / 100.0) / sin(M_PI * (double)m / 3200.0));
        }
        for (int sidx = 1; sidx < 7; sidx++) for (int q = 0; q <= 25; q++) {
            const double a = 2.0 * M_PI * (double)((q * sidx) % 100) / 100.0;
            cpx v = make_float2((float)cos(a), (float)sin(a));
            if (q == 0) v = make_float2(1.0f, 0.0f);
            cs[(sidx - 1) * 26 + q] = v;
        }
        for (int k = -150; k < 850; k++) {
            const int r = ((k % 100) + 100) % 100, j = (k - r) / 100;
            const double a = M_PI * (31.0 * (double)r - 100.0 * (double)j) / 3200.0;
            g1000[k + 150] = make_float2((float)cos(a), (float)sin(a));
        }
        std::vector<cpx> tw100(100);
        for (int m = 0; m < 100; m++) { const double a = -2.0 * M_PI * (double)m / 100.0; tw100[m] = make_float2((float)cos(a), (float)sin(a)); }
        rc |= upload(h, &h->T.K32, k32); rc |= upload(h, &h->T.CS100, cs); rc |= upload(h, &h->T.G1000, g1000); rc |= upload(h, &h->T.TW100, tw100);
    }
    if (rc) { g_create_err = h->err; ft8rx_destroy(h); return -2; }
    // LDPC tables
    bool ok = true;
    ok &= hipMemcpyToSymbol(HIP_SYMBOL(d_CHK_N), FT8_CHK_N, sizeof(FT8_CHK_N)) == hipSuccess;
    ok &= hipMemcpyToSymbol(HIP_SYMBOL(d_CHK_V), FT8_CHK_V, sizeof(FT8_CHK_V)) == hipSuccess;
    {   // k_bp's edge slots, renumbered slot-major (round 6): edge j of check c lives at 83 j + c (j < 6), the seventh edge of the 24
        // degree-7 checks (c >= 59) at 498 + (c - 59).  The check-parallel products then read CONSECUTIVE words, tl[c + 83 j] -- with
        // the generated row-major numbering (check c's edges at 6 c .. 6 c + 5) lane c read word 6 c + j: 16 banks, four lanes each.
        // Only the names of the slots change: products still run over j ascending, a variable's three deltas are still added in
        // the order of FT8_VAR_E (ascending ROW-MAJOR edge number = np.add.at's order, decoders.py:150).
        static uint8_t ev[522], ec[522]; static uint16_t ve[174][3];
        auto slot = [](int e) { const int c = FT8_EDGE_C[e], j = e - FT8_CHK_E0[c]; return j < 6 ? 83 * j + c : 498 + (c - 59); };
        for (int e = 0; e < 522; e++) { ev[slot(e)] = FT8_EDGE_V[e]; ec[slot(e)] = FT8_EDGE_C[e]; }
        for (int v = 0; v < 174; v++) for (int k = 0; k < 3; k++) ve[v][k] = (uint16_t)slot(FT8_VAR_E[v][k]);
        ok &= hipMemcpyToSymbol(HIP_SYMBOL(d_EDGE_V), ev, sizeof(ev)) == hipSuccess;
        ok &= hipMemcpyToSymbol(HIP_SYMBOL(d_EDGE_C), ec, sizeof(ec)) == hipSuccess;
        ok &= hipMemcpyToSymbol(HIP_SYMBOL(d_VAR_E), ve, sizeof(ve)) == hipSuccess;
    }
    ok &= hipMemcpyToSymbol(HIP_SYMBOL(d_G0), FT8_G0, sizeof(FT8_G0)) == hipSuccess;
    {   // per-check membership masks (3 x 64 bits), so a lane gets its two checks' masks with 6 coalesced loads
        static uint64_t cm[128][3];
        memset(cm, 0, sizeof(cm));
        for (int c = 0; c < 83; c++) for (int j = 0; j < FT8_CHK_N[c]; j++) { int v = FT8_CHK_V[c][j]; cm[c][v >> 6] |= 1ull << (v & 63); }
        ok &= hipMemcpyToSymbol(HIP_SYMBOL(d_CHK_MASK), cm, sizeof(cm)) == hipSuccess;
    }
    {   // CRC-14 syndromes of the 77 unit messages (bit-serial definition, decoders.py:123-129)
        uint16_t syn[77];
        for (int pos = 0; pos < 77; pos++) syn[pos] = (uint16_t)hostmsg::crc14_serial(pos < 64 ? (1ull << pos) : 0ull, pos >= 64 ? (1ull << (pos - 64)) : 0ull);
        ok &= hipMemcpyToSymbol(HIP_SYMBOL(d_CRC_SYN), syn, sizeof(syn)) == hipSuccess;
    }
    {   // OSD: G0 column-wise (91 row bits per column) and the byte-wise CRC syndrome table of the 91-bit word
        static uint32_t g0t[192][3];
        memset(g0t, 0, sizeof(g0t));
        for (int r = 0; r < 91; r++) for (int v = 0; v < 174; v++) if ((FT8_G0[r][v >> 6] >> (v & 63)) & 1ull) g0t[v][r >> 5] |= 1u << (r & 31);
        ok &= hipMemcpyToSymbol(HIP_SYMBOL(d_G0T), g0t, sizeof(g0t)) == hipSuccess;
        uint16_t syn91[96]; memset(syn91, 0, sizeof(syn91));
        for (int v = 0; v < 91; v++) {
            if (v < 77) { const int pos = 76 - v; syn91[v] = (uint16_t)hostmsg::crc14_serial(pos < 64 ? (1ull << pos) : 0ull, pos >= 64 ? (1ull << (pos - 64)) : 0ull); }
            else syn91[v] = (uint16_t)(1u << (13 - (v - 77)));
        }
        uint32_t synm[14][3]; memset(synm, 0, sizeof(synm));
        for (int v = 0; v < 91; v++) for (int k = 0; k < 14; k++) if ((syn91[v] >> k) & 1) synm[k][v >> 5] |= 1u << (v & 31);
        ok &= hipMemcpyToSymbol(HIP_SYMBOL(d_SYNM), synm, sizeof(synm)) == hipSuccess;
        static uint16_t ct[12][256];
        for (int b = 0; b < 12; b++) for (int x = 0; x < 256; x++) { uint16_t a = 0; for (int t = 0; t < 8; t++) if ((x >> t) & 1) a ^= syn91[8 * b + t]; ct[b][x] = a; }
        ok &= hipMemcpyToSymbol(HIP_SYMBOL(d_CRC_T), ct, sizeof(ct)) == hipSuccess;
        {   // np.argsort of an all-NaN vector (kernels/osd.hpp: d_NANPERM)
            float xn[174]; int perm[176], stk[64]; uint8_t p8[192];
            for (int i = 0; i < 174; i++) xn[i] = NAN;
            osd_std_sort_withnan(xn, perm, stk);
            memset(p8, 0, sizeof(p8));
            for (int i = 0; i < 174; i++) p8[i] = (uint8_t)perm[i];
            ok &= hipMemcpyToSymbol(HIP_SYMBOL(d_NANPERM), p8, sizeof(p8)) == hipSuccess;
        }
        const std::vector<uint32_t> tr = osd_trial_table(cfg->osd_single, cfg->osd_double, cfg->osd_triple);
        h->n_trials = (int)tr.size();
        if (upload(h, &h->d_trials, tr)) ok = false;
    }
    if (!ok) { set_err(nullptr, "ft8rx_create: device table upload failed"); ft8rx_destroy(h); return -2; }
    if (sync_lds_bytes(*cfg) > 65536) {
        ok &= hipFuncSetAttribute((const void*)k_sync, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sync_lds_bytes(*cfg)) == hipSuccess;
        ok &= hipFuncSetAttribute((const void*)k_sync_acc, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sync_lds_bytes(*cfg)) == hipSuccess;
    }
    if (!ok) { set_err(nullptr, "ft8rx_create: cannot reserve LDS for k_sync"); ft8rx_destroy(h); return -2; }
    // the never-written grid row 0 (receiver.py:240)
    int nfill = (int)B * FT8RX_GRID_COLS;
    k_fill_row0<<<(nfill + 255) / 256, 256, 0, h->stream>>>(h->d_grid, (int)B);
    if (hipStreamSynchronize(h->stream) != hipSuccess) { set_err(nullptr, "ft8rx_create: init kernel failed"); ft8rx_destroy(h); return -2; }
    bool okc = true;
    for (int i = 0; i < 24; i++) { hipEvent_t e = nullptr; okc = okc && hipEventCreate(&e) == hipSuccess; if (e) h->pev.push_back(e); }
    // (the chunk streams h->sub[] are created on first use, launch_batch: every stream that exists competes for one of the
    // runtime's four hardware queues, see the note there)
    // (measurement aid: FT8RX_SUBS_FIRST=n creates n chunk streams BEFORE the copy streams, i.e. gives them hardware queues of their own
    // -- for A/B runs of three or four chunk streams, profiles/r06_notes.md)
    if (const char* e = getenv("FT8RX_SUBS_FIRST")) for (int i = 0; i < atoi(e) && i < 8; i++) pool_sub(device, i);
    okc = okc && (h->copy_s = pool_stream(device, &StreamPool::copy, false)) != nullptr;
    okc = okc && (h->h2d_s = pool_stream(device, &StreamPool::h2d, false)) != nullptr;
    for (int i = 0; i < 16; i++) okc = okc && hipEventCreateWithFlags(&h->ev_chunk[i], hipEventDisableTiming) == hipSuccess;
    okc = okc && hipEventCreateWithFlags(&h->ev_fork, hipEventDisableTiming) == hipSuccess;
    for (int k = 0; k < 2; k++) {
        okc = okc && hipEventCreateWithFlags(&h->ev_comp[k], hipEventDisableTiming) == hipSuccess;
        okc = okc && hipEventCreateWithFlags(&h->ev_done[k], hipEventDisableTiming) == hipSuccess;
    }
    if (!okc) { set_err(nullptr, "ft8rx_create: cannot create HIP streams/events"); ft8rx_destroy(h); return -2; }
    *out = h;
    return 0;
}

int ft8rx_set_profiling(ft8rx_handle* h, int on) { if (!h) return -1; h->profiling = on != 0; return 0; }

int ft8rx_get_stage_times(ft8rx_handle* h, int* n, const char** names, float* ms) {
    if (!h || !n) return -1;
    *n = h->n_stage;
    for (int i = 0; i < h->n_stage; i++) { if (names) names[i] = h->pnames[i]; if (ms) ms[i] = h->stage_ms[i]; }
    return 0;
}


// the kernel chain for frames [f0, f0+B) on stream s (all buffers are frame-major, so a chunk is a pointer offset)
#ifndef LADDER_GRID_CAP
#define LADDER_GRID_CAP (4 * 256 * 32)
#endif

// ladder kernels launch a bounded grid that strides over their work list (a few items per block at most): enough blocks to fill the
// chip four times over, never more than there can be items
static int ladder_grid(int max_items) { const int cap = LADDER_GRID_CAP; return max_items < cap ? max_items : cap; }

// the sync search of frames [.., B) over the configured h0 range, in windows of SYNC_WIN offsets (one launch for any range up to 14 s)
static void launch_sync(const float* grid, float* bs, int32_t* bh, const ft8rx_config& c, int B, hipStream_t s) {
    const int ntile = (c.f0_hi - c.f0_lo + 15) / 16;
    for (int lo = c.h0_lo; lo < c.h0_hi; lo += SYNC_WIN) {
        ft8rx_config w = c;
        w.h0_lo = lo; w.h0_hi = lo + SYNC_WIN < c.h0_hi ? lo + SYNC_WIN : c.h0_hi;
        if (lo == c.h0_lo) k_sync<<<dim3(ntile, B), 256, sync_lds_bytes(c), s>>>(grid, bs, bh, w);
        else k_sync_acc<<<dim3(ntile, B), 256, sync_lds_bytes(c), s>>>(grid, bs, bh, w);
    }
}

static void enqueue_chain(ft8rx_handle* h, const int16_t* d_audio, int f0, int B, hipStream_t s, bool prof, int slot, int chunk) {
    const ft8rx_config& c = h->cfg;
    const size_t F = (size_t)f0;
    const int16_t* audio = d_audio + F * FT8RX_NSAMP;
    float* grid = h->d_grid + F * FT8RX_GRID_ROWS * FT8RX_GRID_COLS;
    float* bs = h->d_best_score + F * NF0MAX; int32_t* bh = h->d_best_h0 + F * NF0MAX;
    ft8rx_record* rec = h->s_rec[slot] + F * MAXC; int32_t* ncand = h->s_ncand[slot] + F;
    float* llr0 = h->d_llr0 + F * MAXC * 174; float* saved = h->d_saved + F * MAXC * 5 * 174;
    Att* att0 = h->d_att0 + F * MAXC * 5; Att* attG = h->d_attG + F * MAXC * 2; Att* attB = h->d_attB + F * MAXC * 5; Att* attO = h->d_attO + F * MAXC * 10;
    cpx* A = reinterpret_cast<cpx*>(grid); cpx* spec = A + (size_t)B * 96000;      // inside this chunk's (by then dead) grid region, see ft8rx_create
    ft8rx_event* ev = h->s_ev[slot] + F * FT8RX_EVENT_CAP; int32_t* evc = h->s_evcount[slot] + F;
#define STAGE(name) do { if (prof) { hipEventRecord(h->pev[h->pnames.size()], s); h->pnames.push_back(name); } } while (0)
    int32_t* wc = h->d_wcount + WL_N * chunk;                      // (evc and wc are zeroed by k_topk)
    WorkList wl[WL_N];
    for (int i = 0; i < WL_N; i++) { wl[i].items = h->d_work[i] + F * MAXC * WL_MULT(i); wl[i].count = wc + i; }
    STAGE("spectrogram");
    ft8rx_ilp_spectrogram(B, s, audio, grid, h->T);
    STAGE("sync");
    launch_sync(grid, bs, bh, c, B, s);
    STAGE("topk");
    k_topk<<<B, 1024, 0, s>>>(bs, bh, rec, ncand, c, evc, wc, h->use_mask ? h->d_colmask + F * NF0MAX : nullptr);
    STAGE("grid_llr");
    k_grid_llr<<<XCD_GRID(B, c.max_cands), 64, 0, s>>>(grid, rec, ncand, llr0, c, nullptr, nullptr, nullptr, att0, ev, evc, B);
    k_worklist_att<<<(B * MAXC * 5 + 255) / 256, 256, 0, s>>>(rec, ncand, att0, B, wl[WL_BP0]);
    STAGE("bp_grid");
    k_bp<<<ladder_grid(B * c.max_cands * 5), 64, 0, s>>>(0, llr0, rec, ncand, nullptr, att0, nullptr, ev, evc, c, c.bp_nc0_a, c.bp_iters_a, wl[WL_BP0], 0, 5);
    STAGE("select0");
    k_select0<<<(B * MAXC + 255) / 256, 256, 0, s>>>(rec, ncand, att0, B, wl[WL_FINE]);
    STAGE("cycle_fft");
    k_cyc_a<<<dim3(40, B), 256, 0, s>>>(audio, A, h->T);
    k_cyc_bc<<<dim3(CYC_BC_GRID, B), 256, 0, s>>>(A, spec, h->T);
    STAGE("fine");
    ft8rx_ilp_fine(ladder_grid(B * c.max_cands), s, spec, rec, ncand, llr0, h->T, c, nullptr, nullptr, nullptr, nullptr, wl[WL_FINE]);
    if (c.h0_lo < FT8RX_MIN_H0_FD || c.h0_hi > FT8RX_MAX_H0_FD + 1)      // a search_time_range beyond -6.1 .. +8.3 s: the candidates k_fine leaves out
        k_fine_td<<<ladder_grid(B * c.max_cands), FINE_NT, 0, s>>>(spec, rec, ncand, llr0, h->T, c, nullptr, nullptr, nullptr, nullptr, wl[WL_FINE]);
    k_worklist<<<(B * MAXC + 255) / 256, 256, 0, s>>>(rec, ncand, B, wl[WL_BP1]);
    STAGE("bp_fine");
    // fine-stage BP: in ladder order (three launches; decided candidates drop out), or -- ft8rx_set_ladder_mode(h, 1), for small
    // batches where latency matters more than work -- all five variants in one launch: one dependent BP instead of three, same
    // records and messages (the event log then also holds entries of attempts the ladder would not have reached)
    if (h->ladder_mode == 0) {
        k_bp<<<B * c.max_cands, 64, 0, s>>>(1, llr0, rec, ncand, attG, attB, saved, ev, evc, c, c.bp_nc0_b, c.bp_iters_b, wl[WL_BP1], 0, 1);
        k_select1<<<(B * MAXC + 255) / 256, 256, 0, s>>>(0, rec, ncand, attG, attB, B, c, wl[WL_BP1B]);
        k_bp<<<B * c.max_cands, 64, 0, s>>>(1, llr0, rec, ncand, attG, attB, saved, ev, evc, c, c.bp_nc0_b, c.bp_iters_b, wl[WL_BP1B], 1, 1);
        k_select1<<<(B * MAXC + 255) / 256, 256, 0, s>>>(1, rec, ncand, attG, attB, B, c, wl[WL_BP1C]);
        k_bp<<<B * c.max_cands * 3, 64, 0, s>>>(1, llr0, rec, ncand, attG, attB, saved, ev, evc, c, c.bp_nc0_b, c.bp_iters_b, wl[WL_BP1C], 2, 3);
        STAGE("select1");
        k_select1<<<(B * MAXC + 255) / 256, 256, 0, s>>>(2, rec, ncand, attG, attB, B, c, wl[WL_OSD]);
    } else {
        k_bp<<<B * c.max_cands * 5, 64, 0, s>>>(1, llr0, rec, ncand, attG, attB, saved, ev, evc, c, c.bp_nc0_b, c.bp_iters_b, wl[WL_BP1], 0, 5);
        STAGE("select1");
        k_select1<<<(B * MAXC + 255) / 256, 256, 0, s>>>(3, rec, ncand, attG, attB, B, c, wl[WL_OSD]);
    }
    STAGE("osd");
    const bool osd_wide = osd_nflip(c.osd_single, c.osd_triple) > OSD_FLIPS_A;
    (osd_wide ? k_osd_wide : k_osd)<<<ladder_grid(B * c.max_cands * 10), 64, 0, s>>>(
        0, llr0, saved, attB, rec, ncand, attO, ev, evc, h->d_trials, h->n_trials, osd_nflip(c.osd_single, c.osd_triple), c.osd_max_hd, wl[WL_OSD], wl[WL_OSDNAN]);
    // attempts on vectors with a NaN (a NaN-poisoned BP output): the reference's numpy orders those with std::sort -- a kernel of their own
    (osd_wide ? k_osd_nan_wide : k_osd_nan)<<<OSD_NAN_GRID, 64, 0, s>>>(
        0, llr0, saved, attB, rec, ncand, attO, ev, evc, h->d_trials, h->n_trials, osd_nflip(c.osd_single, c.osd_triple), c.osd_max_hd, wl[WL_OSDNAN]);
    STAGE("select2");
    k_select2<<<(B * MAXC + 255) / 256, 256, 0, s>>>(rec, ncand, attO, B);
    if (prof) hipEventRecord(h->pev[h->pnames.size()], s);
#undef STAGE
}

// A chunk's frames go through the WHOLE chain in cache-sized sub-batches, one after the other on the chunk's stream: a sub-batch's dB
// grid (1.47 MB per frame) is then still in L2 / the 256 MB MALL when k_sync, k_topk and k_grid_llr read it, and the four-step scratch
// and cycle spectrum that later overlay it are too when k_cyc_bc / k_fine read them -- a 4096-frame chunk's grid is 6 GB and every
// stage of it came from HBM (k_spectrogram 0.36 -> 0.42 ms per 256 frames, profiles/archive/r04_batch_sweep.txt).  Every workspace is indexed
// by frame and the chunk's work-list counters are re-zeroed by each chain's k_topk, so consecutive sub-batches on one stream need
// nothing but stream order.  Results do not depend on the partition (tests/test_gpu_parity.py: test_subbatch_partition_invariance).
static void enqueue_chunk(ft8rx_handle* h, const int16_t* d_audio, int f0, int n, hipStream_t s, int slot, int chunk) {
    const int sub = (h->sub_frames > 0 && h->sub_frames < n) ? h->sub_frames : n;
    for (int o = 0; o < n; o += sub) enqueue_chain(h, d_audio, f0 + o, (o + sub <= n) ? sub : n - o, s, false, slot, chunk);
}

// Launch one batch: audio either resident on the device (host_audio == nullptr) or copied from the host in chunks.
//   profiling mode / small batches: one chain on the main stream (per-stage events bracket whole-batch launches);
//   otherwise the batch is cut into chunks whose chains overlap on the sub-streams -- the ladder kernels (BP, OSD, fine sync)
//   are latency bound, so chunks fill each other's stalls.  Host audio uses twice as many chunks; their copies run in order on
//   the dedicated copy stream (never queued behind kernels, so a pageable-memory copy blocks the host only for its own
//   duration) and each chunk's chain waits for its copy's event.
// When the kernels are done the copy stream moves the slot's results into the page-locked host buffers.
// pipelined = true (ft8rx_enqueue_batch_host): the audio is staged in the buffer of this batch's result slot and its copies run on
// their own stream, gated only by the previous user of that staging buffer (two batches ago), so they overlap the kernels of
// the batch before.
static int launch_batch(ft8rx_handle* h, const int16_t* d_audio, const int16_t* host_audio, int B, bool pipelined = false) {
    HIPCHK(h, hipSetDevice(h->device));
    h->pnames.clear();
    if (h->inflight == 2) { h->slot_fetch ^= 1; h->inflight = 1; }          // the oldest unfetched batch is dropped
    const int slot = h->slot_enq;
    if (host_audio && need_staging(h)) return -2;
    int16_t* stage = h->d_audio;
    hipStream_t cs = h->copy_s;
    if (pipelined) {
        if (slot == 1 && !h->d_audio2 && dalloc(h, &h->d_audio2, (size_t)h->max_frames * FT8RX_NSAMP)) return -2;
        stage = slot ? h->d_audio2 : h->d_audio;
        d_audio = stage;
        cs = h->h2d_s;
        HIPCHK(h, hipStreamWaitEvent(cs, h->ev_comp[slot], 0));             // the kernels that last read this staging buffer are done
    }
    // synchronous host entry: twice as many chunks, so that the first kernels start after 1/8 of the copy; the pipelined entry's copies
    // already overlap the previous batch, so it keeps the 4 larger chunks of the device-resident path
    int nc = h->profiling ? 1 : ((host_audio && !pipelined) ? 2 * h->n_streams : h->n_streams);
    if (nc > B / 8) nc = B / 8;
    // Chunk k runs on stream k % n_streams, where stream 0 IS the handle's main stream and streams 1.. are the (lazily created)
    // sub-streams.  The HIP runtime maps all streams of a process onto four hardware queues, and commands of streams that share
    // one execute in submission order: with a main stream that only forks and joins plus two chunk streams plus the two copy
    // streams (five), the H2D stream shared a queue with a chunk stream and its event markers waited behind that chunk's kernels --
    // in the rocprofv3 trace the second H2D chunk of the pipelined host entry started only when the previous batch's last kernel
    // had finished (profiles/archive/r03_notes.md).  Four streams in use = a queue each.
    const int ns = h->n_streams;
    for (int i = 1; i < ns && nc > 1; i++) if (!h->sub[i - 1]) {
        if (!(h->sub[i - 1] = pool_sub(h->device, i - 1))) { set_err(h, "cannot create a chunk stream"); return -2; }
        HIPCHK(h, hipEventCreateWithFlags(&h->ev_join[i - 1], hipEventDisableTiming));
    }
    auto chunk_stream = [&](int k) { const int i = k % ns; return i == 0 ? h->stream : h->sub[i - 1]; };
    // Free-running chunk streams: with the audio resident (or staged by the pipelined host entry) and one chunk per stream, chunk i of
    // batch k+1 simply follows chunk i of batch k on stream i.  Every workspace is indexed by frame, so a stream only ever touches
    // its own frame range and in-stream order is all the ordering the chains need; only the result copy waits for all of them.
    // Without the per-batch fork / join a stream that finishes its half early starts on the next batch while the other one is still
    // in its tail.  Anything else that uses the workspaces (stage entry points, the synchronous entry, a different partition)
    // goes through quiesce() / need_barrier.
    // (A/B on one box, profiles/archive/r03_notes.md: 48 498 -> 49 001 frames/s end to end, 48 167 -> 48 894 including H2D.)
    const bool free_run = !h->profiling && nc > 1 && nc == ns && !(host_audio && !pipelined);
    if (free_run) {
        for (int i = 0; i < ns; i++) for (int k = 0; k < 2; k++)
            if (!h->ev_cdone[k][i]) HIPCHK(h, hipEventCreateWithFlags(&h->ev_cdone[k][i], hipEventDisableTiming));
        const int per = (B + nc - 1) / nc;
        if (h->need_barrier || h->part_B != B || h->part_n != nc) {
            // first batch of a run (or another partition): everything issued so far, on any stream of this handle, first
            if (h->free_running) HIPCHK(h, hipStreamWaitEvent(h->stream, h->ev_comp[h->last_slot], 0));
            HIPCHK(h, hipEventRecord(h->ev_fork, h->stream));
            for (int i = 1; i < ns; i++) HIPCHK(h, hipStreamWaitEvent(h->sub[i - 1], h->ev_fork, 0));
            h->need_barrier = false; h->part_B = B; h->part_n = nc;
        }
        if (pipelined) {
            HIPCHK(h, hipMemcpyAsync(stage, host_audio, sizeof(int16_t) * (size_t)B * FT8RX_NSAMP, hipMemcpyHostToDevice, cs));
            HIPCHK(h, hipEventRecord(h->ev_chunk[0], cs));
        }
        for (int k = 0; k < nc; k++) {
            const int f0 = k * per, n = (f0 + per <= B) ? per : B - f0;
            hipStream_t s = chunk_stream(k);
            HIPCHK(h, hipStreamWaitEvent(s, h->ev_done[slot], 0));          // the slot's previous results have left the device
            if (pipelined) HIPCHK(h, hipStreamWaitEvent(s, h->ev_chunk[0], 0));
            if (n > 0) enqueue_chunk(h, d_audio, f0, n, s, slot, k);
            HIPCHK(h, hipEventRecord(h->ev_cdone[slot][k], s));
            HIPCHK(h, hipStreamWaitEvent(h->copy_s, h->ev_cdone[slot][k], 0));
        }
        h->free_running = true;
    } else {
    { const int q = quiesce(h, true); if (q) return q; }
    HIPCHK(h, hipStreamWaitEvent(h->stream, h->ev_done[slot], 0));          // the slot's previous results have left the device
    if (nc <= 1) {
        if (host_audio) HIPCHK(h, hipMemcpyAsync(stage, host_audio, sizeof(int16_t) * (size_t)B * FT8RX_NSAMP, hipMemcpyHostToDevice, h->stream));
        if (h->profiling) enqueue_chain(h, d_audio, 0, B, h->stream, true, slot, 0);      // per-stage events bracket whole-batch launches
        else enqueue_chunk(h, d_audio, 0, B, h->stream, slot, 0);
    } else {
        HIPCHK(h, hipEventRecord(h->ev_fork, h->stream));
        if (host_audio && !pipelined) HIPCHK(h, hipStreamWaitEvent(cs, h->ev_fork, 0));
        for (int i = 1; i < ns; i++) HIPCHK(h, hipStreamWaitEvent(h->sub[i - 1], h->ev_fork, 0));
        const int per = (B + nc - 1) / nc;
        // chunk boundaries: equal parts, except for the synchronous host entry, whose first kernels can only start when the first
        // chunk's audio has crossed PCIe -- there the chunks grow geometrically (B/8, B/8, B/4, B/2 for four): the first copy is
        // half as long and the large chunks, which run most efficiently, come last (38.8 k -> 40.5 k frames/s for 256-frame calls;
        // other layouts -- 16/48/64/128, three streams, five or six chunks -- all land between 38 k and 41.5 k)
        int cb[17];
        for (int k = 0; k <= nc; k++) cb[k] = (k * per < B) ? k * per : B;
        if (host_audio && !pipelined && nc >= 3 && B >= 8 * nc) {
            int left = B;
            for (int k = nc - 1; k >= 1; k--) { const int n = left / 2; cb[k] = left - n; left -= n; }
            cb[0] = 0; cb[nc] = B;
        }
        if (pipelined) {
            // ONE copy for the whole batch and one event: it has the whole previous batch to hide behind, and a copy stream with
            // event markers BETWEEN its copies can stall on them (markers are queue packets: in a hardware queue shared with a
            // chunk stream they wait their turn behind that chunk's kernels -- the second of two chunk copies then started only
            // at the end of the previous batch, rocprofv3 trace in profiles/archive/r03_notes.md)
            HIPCHK(h, hipMemcpyAsync(stage, host_audio, sizeof(int16_t) * (size_t)B * FT8RX_NSAMP, hipMemcpyHostToDevice, cs));
            HIPCHK(h, hipEventRecord(h->ev_chunk[0], cs));
            for (int i = 0; i < ns; i++) HIPCHK(h, hipStreamWaitEvent(chunk_stream(i), h->ev_chunk[0], 0));
        }
        for (int k = 0; k < nc; k++) {
            const int f0 = cb[k], n = cb[k + 1] - cb[k];
            if (n <= 0) continue;
            hipStream_t s = chunk_stream(k);
            if (host_audio && !pipelined) {
                HIPCHK(h, hipMemcpyAsync(stage + (size_t)f0 * FT8RX_NSAMP, host_audio + (size_t)f0 * FT8RX_NSAMP,
                                         sizeof(int16_t) * (size_t)n * FT8RX_NSAMP, hipMemcpyHostToDevice, cs));
                HIPCHK(h, hipEventRecord(h->ev_chunk[k], cs));
                HIPCHK(h, hipStreamWaitEvent(s, h->ev_chunk[k], 0));
            }
            enqueue_chunk(h, d_audio, f0, n, s, slot, k);
        }
        for (int i = 1; i < ns; i++) {
            HIPCHK(h, hipEventRecord(h->ev_join[i - 1], h->sub[i - 1]));
            HIPCHK(h, hipStreamWaitEvent(h->stream, h->ev_join[i - 1], 0));
        }
    }
    }
    HIPCHK(h, hipGetLastError());
    // results -> page-locked host buffers on the copy stream (overlaps the next batch, which computes into the other slot).
    // fin = the stream on which "all kernels of this batch are done" is known: the main stream after its joins, or -- free-running
    // chunks -- the copy stream, which has just been made to wait for every chunk
    const int mc = h->cfg.max_cands;
    hipStream_t fin = free_run ? h->copy_s : h->stream;
    if (h->d_evpacked[slot] && (size_t)B * FT8RX_EVENT_CAP * sizeof(ft8rx_event) > EV_EAGER_BYTES) {      // pack the event log (see the copy below)
        k_ev_scan<<<1, 1024, 0, fin>>>(h->s_evcount[slot], B, h->d_evoffs[slot]);
        k_ev_compact<<<B, 64, 0, fin>>>(h->s_ev[slot], h->s_evcount[slot], h->d_evoffs[slot], h->d_evpacked[slot]);
    }
    h->slot_packed[slot] = h->pk_buf[slot] != nullptr;
    if (h->slot_packed[slot]) {                              // packed results for a gather: header | frame table | kept records | used events
        if (h->pk_fence[slot]) { HIPCHK(h, hipStreamWaitEvent(fin, h->pk_fence[slot], 0)); h->pk_fence[slot] = nullptr; }      // the consumer's asynchronous read of this buffer
        k_pack_count<<<B, 256, 0, fin>>>(h->s_rec[slot], h->s_ncand[slot], h->s_ev[slot], h->s_evcount[slot], h->d_pkneed, h->d_pknrec);
        k_pack_scan<<<1, 1024, 0, fin>>>(h->d_pknrec, h->s_ncand[slot], h->s_evcount[slot], B, mc, (unsigned long long)h->pk_cap, h->pk_buf[slot], h->d_pkhdr[slot]);
        k_pack_write<<<B, 256, 0, fin>>>(h->s_rec[slot], h->s_ev[slot], h->d_pkneed, B, h->pk_buf[slot]);
    }
    HIPCHK(h, hipEventRecord(h->ev_comp[slot], fin));
    if (!free_run) HIPCHK(h, hipStreamWaitEvent(h->copy_s, h->ev_comp[slot], 0));
    HIPCHK(h, hipMemcpyAsync(h->h_cnt[slot], h->s_ncand[slot], sizeof(int32_t) * B, hipMemcpyDeviceToHost, h->copy_s));
    HIPCHK(h, hipMemcpy2DAsync(h->h_rec[slot], sizeof(ft8rx_record) * mc, h->s_rec[slot], sizeof(ft8rx_record) * MAXC,
                               sizeof(ft8rx_record) * mc, B, hipMemcpyDeviceToHost, h->copy_s));
    HIPCHK(h, hipMemcpyAsync(h->h_evc[slot], h->s_evcount[slot], sizeof(int32_t) * B, hipMemcpyDeviceToHost, h->copy_s));
    // The event log is [B][FT8RX_EVENT_CAP] x 24 B = 12 KB per frame of which a frame typically uses a tenth (config 1: ~40 events).
    // Small batches copy it whole (the latency case).  Large ones were packed above: k_ev_compact has already written the used
    // entries -- and nothing else -- into page-locked host memory: ~8 MB instead of 100 MB per 8192-frame shard on the host link.
    // (A copy-engine transfer of the packed run is not an option: an asynchronous device-to-host copy of a few hundred KB takes the
    // runtime's shader-copy path, which queued behind the next batch's kernels and cost 0.5 ms per 256-frame step.)
    h->slot_evpending[slot] = h->d_evpacked[slot] && (size_t)B * FT8RX_EVENT_CAP * sizeof(ft8rx_event) > EV_EAGER_BYTES;
    if (!h->slot_evpending[slot])
        HIPCHK(h, hipMemcpyAsync(h->h_ev[slot], h->s_ev[slot], sizeof(ft8rx_event) * (size_t)B * FT8RX_EVENT_CAP, hipMemcpyDeviceToHost, h->copy_s));
    HIPCHK(h, hipEventRecord(h->ev_done[slot], h->copy_s));
    h->slot_B[slot] = B; h->last_slot = slot; h->slot_enq ^= 1; h->inflight++;
    return 0;
}

// Large batches: the packed event log (k_ev_compact, written by the GPU into page-locked memory) -> the [frame][FT8RX_EVENT_CAP] rows
// of h_ev that the fetch functions hand out.  The per-frame counts have arrived, so the host knows each frame's offset in the
// packed run (the same prefix sum k_ev_scan made).  Rows beyond a frame's count are not written.
static int fetch_events(ft8rx_handle* h, int slot) {
    if (!h->slot_evpending[slot]) return 0;
    const int B = h->slot_B[slot];
    size_t off = 0;
    for (int f = 0; f < B; f++) {
        int c = h->h_evc[slot][f]; c = c > FT8RX_EVENT_CAP ? FT8RX_EVENT_CAP : (c < 0 ? 0 : c);
        if (c) memcpy(h->h_ev[slot] + (size_t)f * FT8RX_EVENT_CAP, h->h_evpacked[slot] + off, sizeof(ft8rx_event) * (size_t)c);
        off += (size_t)c;
    }
    h->slot_evpending[slot] = false;
    return 0;
}

int ft8rx_enqueue_batch(ft8rx_handle* h, const int16_t* d_audio, int B) {
    if (!h || !d_audio) return -1;
    if (B < 1 || B > h->max_frames) { set_err(h, "ft8rx_enqueue_batch: n_frames %d outside [1, %d]", B, h->max_frames); return -1; }
    return launch_batch(h, d_audio, nullptr, B);
}

int ft8rx_enqueue_batch_host(ft8rx_handle* h, const int16_t* audio, int B) {
    if (!h || !audio) return -1;
    if (B < 1 || B > h->max_frames) { set_err(h, "ft8rx_enqueue_batch_host: n_frames %d outside [1, %d]", B, h->max_frames); return -1; }
    return launch_batch(h, nullptr, audio, B, true);
}

int ft8rx_set_streams(ft8rx_handle* h, int n) { if (!h || n < 1 || n > 8) return -1; h->n_streams = n; return 0; }
int ft8rx_set_subbatch(ft8rx_handle* h, int frames) { if (!h || frames < 0) return -1; h->sub_frames = frames; return 0; }
int ft8rx_set_ladder_mode(ft8rx_handle* h, int mode) { if (!h || mode < 0 || mode > 1) return -1; h->ladder_mode = mode; return 0; }

int ft8rx_set_search_mask(ft8rx_handle* h, const uint8_t* mask, int n_frames) {
    if (!h) return -1;
    ENTER(h);                                   // batches in flight were enqueued under the previous setting
    if (!mask) { h->use_mask = false; return 0; }
    if (n_frames < 1 || n_frames > h->max_frames) { set_err(h, "ft8rx_set_search_mask: n_frames %d outside [1, %d]", n_frames, h->max_frames); return -1; }
    if (!h->d_colmask && dalloc(h, &h->d_colmask, (size_t)h->max_frames * NF0MAX)) return -2;
    const int nf0 = h->cfg.f0_hi - h->cfg.f0_lo;
    HIPCHK(h, hipMemsetAsync(h->d_colmask, 0, (size_t)h->max_frames * NF0MAX, h->stream));
    HIPCHK(h, hipMemcpy2DAsync(h->d_colmask, NF0MAX, mask, (size_t)nf0, (size_t)nf0, n_frames, hipMemcpyHostToDevice, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    h->use_mask = true;
    return 0;
}

int ft8rx_sync(ft8rx_handle* h) {
    if (!h) return -1;
    HIPCHK(h, hipSetDevice(h->device));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    HIPCHK(h, hipStreamSynchronize(h->copy_s));
    if (h->profiling && !h->pnames.empty()) {
        h->n_stage = (int)h->pnames.size();
        for (int i = 0; i < h->n_stage; i++) hipEventElapsedTime(&h->stage_ms[i], h->pev[i], h->pev[i + 1]);
    }
    return 0;
}

int ft8rx_fetch_results(ft8rx_handle* h, int B, ft8rx_record* records, int32_t* counts, ft8rx_event* events, int32_t* event_counts) {
    if (!h || B < 1 || B > h->max_frames) return -1;
    if (h->last_slot < 0) { set_err(h, "ft8rx_fetch_results: nothing has been enqueued"); return -1; }
    HIPCHK(h, hipSetDevice(h->device));
    const int slot = h->inflight ? h->slot_fetch : h->last_slot;      // oldest unfetched batch, else the latest one again
    if (B > h->slot_B[slot]) { set_err(h, "ft8rx_fetch_results: %d frames requested, the batch had %d", B, h->slot_B[slot]); return -1; }
    HIPCHK(h, hipEventSynchronize(h->ev_done[slot]));
    if (events) { const int rc = fetch_events(h, slot); if (rc) return rc; }
    const size_t mc = (size_t)h->cfg.max_cands;
    if (counts) memcpy(counts, h->h_cnt[slot], sizeof(int32_t) * B);
    if (records) memcpy(records, h->h_rec[slot], sizeof(ft8rx_record) * mc * B);
    if (event_counts) memcpy(event_counts, h->h_evc[slot], sizeof(int32_t) * B);
    if (events) for (int f = 0; f < B; f++) {              // the used entries of each frame; the rest of the caller's row is left alone
        int c = h->h_evc[slot][f]; if (c > FT8RX_EVENT_CAP) c = FT8RX_EVENT_CAP;
        if (c > 0) memcpy(events + (size_t)f * FT8RX_EVENT_CAP, h->h_ev[slot] + (size_t)f * FT8RX_EVENT_CAP, sizeof(ft8rx_event) * (size_t)c);
    }
    h->fetched_slot = slot;
    if (h->inflight) { h->slot_fetch ^= 1; h->inflight--; }
    return 0;
}

int ft8rx_fetch_results_view(ft8rx_handle* h, int B, const ft8rx_record** records, const int32_t** counts,
                             const ft8rx_event** events, const int32_t** event_counts) {
    if (!h || B < 1 || B > h->max_frames) return -1;
    if (h->last_slot < 0) { set_err(h, "ft8rx_fetch_results_view: nothing has been enqueued"); return -1; }
    HIPCHK(h, hipSetDevice(h->device));
    const int slot = h->inflight ? h->slot_fetch : h->last_slot;
    if (B > h->slot_B[slot]) { set_err(h, "ft8rx_fetch_results_view: %d frames requested, the batch had %d", B, h->slot_B[slot]); return -1; }
    HIPCHK(h, hipEventSynchronize(h->ev_done[slot]));
    if (events) { const int rc = fetch_events(h, slot); if (rc) return rc; }
    if (records) *records = h->h_rec[slot];
    if (counts) *counts = h->h_cnt[slot];
    if (events) *events = h->h_ev[slot];
    if (event_counts) *event_counts = h->h_evc[slot];
    h->fetched_slot = slot;
    if (h->inflight) { h->slot_fetch ^= 1; h->inflight--; }
    return 0;
}

int ft8rx_set_packed_output(ft8rx_handle* h, void* d_buf0, void* d_buf1, uint64_t cap_bytes) {
    if (!h) return -1;
    ENTER(h);                                   // batches in flight keep the buffers they were enqueued with
    HIPCHK(h, hipStreamSynchronize(h->copy_s));
    h->pk_fence[0] = h->pk_fence[1] = nullptr;
    if (!d_buf0 && !d_buf1) { h->pk_buf[0] = h->pk_buf[1] = nullptr; h->pk_cap = 0; return 0; }
    if (!d_buf0 || !d_buf1 || d_buf0 == d_buf1 || cap_bytes < sizeof(ft8rx_packed_header)) {
        set_err(h, "ft8rx_set_packed_output: two distinct buffers of at least %zu bytes each are needed", sizeof(ft8rx_packed_header)); return -1; }
    void* in[2] = {d_buf0, d_buf1};
    unsigned char* dev[2];
    for (int k = 0; k < 2; k++) {               // page-locked host memory is addressed through its device pointer
        hipPointerAttribute_t at;
        if (hipPointerGetAttributes(&at, in[k]) != hipSuccess) { (void)hipGetLastError(); set_err(h, "ft8rx_set_packed_output: buffer %d is neither device nor page-locked host memory", k); return -1; }
        if (at.type != hipMemoryTypeDevice && at.type != hipMemoryTypeHost && at.type != hipMemoryTypeManaged) {      // e.g. plain malloc memory: the kernels could not write it
            set_err(h, "ft8rx_set_packed_output: buffer %d is ordinary host memory; use device memory or ft8rx_alloc_host", k); return -1; }
        dev[k] = (unsigned char*)(at.type == hipMemoryTypeHost ? at.devicePointer : in[k]);
        if (!dev[k] || ((uintptr_t)dev[k] & 15)) { set_err(h, "ft8rx_set_packed_output: buffer %d is not device-accessible / 16-byte aligned", k); return -1; }
    }
    if (!h->d_pkneed) {
        int rc = dalloc(h, &h->d_pkneed, (size_t)h->max_frames * PK_NW);
        rc |= dalloc(h, &h->d_pknrec, (size_t)h->max_frames);
        if (rc) return -2;
        for (int k = 0; k < 2; k++) {
            if (hipHostMalloc((void**)&h->h_pkhdr[k], sizeof(ft8rx_packed_header), hipHostMallocDefault) != hipSuccess ||
                hipHostGetDevicePointer((void**)&h->d_pkhdr[k], h->h_pkhdr[k], 0) != hipSuccess) { set_err(h, "ft8rx_set_packed_output: page-locked header could not be allocated"); return -2; }
            memset(h->h_pkhdr[k], 0, sizeof(ft8rx_packed_header));
        }
    }
    h->pk_buf[0] = dev[0]; h->pk_buf[1] = dev[1]; h->pk_cap = cap_bytes;
    return 0;
}

int ft8rx_packed_output_fence(ft8rx_handle* h, int which, void* hip_event) {
    if (!h || which < 0 || which > 1) return -1;
    h->pk_fence[which] = (hipEvent_t)hip_event;
    return 0;
}

int ft8rx_packed_results(ft8rx_handle* h, int32_t* which, ft8rx_packed_header* header) {
    if (!h) return -1;
    if (h->fetched_slot < 0 || !h->slot_packed[h->fetched_slot]) { set_err(h, "ft8rx_packed_results: the batch fetched last was enqueued without a packed output (ft8rx_set_packed_output)"); return -1; }
    if (which) *which = h->fetched_slot;
    if (header) *header = *h->h_pkhdr[h->fetched_slot];
    return 0;
}

int ft8rx_results_to_device(ft8rx_handle* h, int B, ft8rx_record* d_records, int32_t* d_counts, ft8rx_event* d_events, int32_t* d_event_counts) {
    if (!h || B < 1 || B > h->max_frames) return -1;
    if (h->last_slot < 0) { set_err(h, "ft8rx_results_to_device: nothing has been enqueued"); return -1; }
    HIPCHK(h, hipSetDevice(h->device));
    const int slot = h->last_slot;
    if (B > h->slot_B[slot]) { set_err(h, "ft8rx_results_to_device: %d frames requested, the batch had %d", B, h->slot_B[slot]); return -1; }
    HIPCHK(h, hipEventSynchronize(h->ev_comp[slot]));                // the batch's kernels are done on every chunk stream (results sit in s_*[slot])
    HIPCHK(h, hipStreamSynchronize(h->stream));
    const int mc = h->cfg.max_cands;
    if (d_counts) HIPCHK(h, hipMemcpyAsync(d_counts, h->s_ncand[slot], sizeof(int32_t) * B, hipMemcpyDeviceToDevice, h->stream));
    if (d_records) HIPCHK(h, hipMemcpy2DAsync(d_records, sizeof(ft8rx_record) * mc, h->s_rec[slot], sizeof(ft8rx_record) * MAXC,
                                              sizeof(ft8rx_record) * mc, B, hipMemcpyDeviceToDevice, h->stream));
    if (d_event_counts) HIPCHK(h, hipMemcpyAsync(d_event_counts, h->s_evcount[slot], sizeof(int32_t) * B, hipMemcpyDeviceToDevice, h->stream));
    if (d_events) HIPCHK(h, hipMemcpyAsync(d_events, h->s_ev[slot], sizeof(ft8rx_event) * (size_t)B * FT8RX_EVENT_CAP, hipMemcpyDeviceToDevice, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    return 0;
}

int ft8rx_decode_batch(ft8rx_handle* h, const int16_t* audio, int B, ft8rx_record* records, int32_t* counts,
                       ft8rx_event* events, int32_t* event_counts) {
    if (!h || !audio) return -1;
    if (B < 1 || B > h->max_frames) { set_err(h, "ft8rx_decode_batch: n_frames %d outside [1, %d]", B, h->max_frames); return -1; }
    h->inflight = 0; h->slot_fetch = h->slot_enq;                     // synchronous entry: nothing older is kept
    if (need_staging(h)) return -2;
    int rc = launch_batch(h, h->d_audio, audio, B);
    if (rc) return rc;
    return ft8rx_fetch_results(h, B, records, counts, events, event_counts);
}

// One call from host audio to rendered messages: ft8rx_decode_batch + ft8rx_package_batch without the intermediate copies (the
// packager reads the handle's page-locked result buffers in place).
int ft8rx_decode_messages(ft8rx_handle* h, const int16_t* audio, int B, ft8rx_message* out, int max_msgs, int32_t* out_counts,
                          int n_threads, ft8rx_hashes* table, int32_t* flags) {
    if (!h || !audio || !out || !out_counts) return -1;
    if (B < 1 || B > h->max_frames || max_msgs < 1) { set_err(h, "ft8rx_decode_messages: bad n_frames / max_msgs"); return -1; }
    h->inflight = 0; h->slot_fetch = h->slot_enq;                     // synchronous entry: nothing older is kept
    if (need_staging(h)) return -2;
    int rc = launch_batch(h, h->d_audio, audio, B);
    if (rc) return rc;
    const ft8rx_record* rec; const int32_t* cnt; const ft8rx_event* ev; const int32_t* evc;
    rc = ft8rx_fetch_results_view(h, B, &rec, &cnt, &ev, &evc);
    if (rc) return rc;
    return hostmsg::package_batch(rec, cnt, ev, evc, B, h->cfg.max_cands, out, max_msgs, out_counts, n_threads, table ? &table->H : nullptr, flags);
}

// ---------------------------------------------------------------------------- stage entry points
#define NEED(p) do { if (!(p)) { set_err(h, "scratch allocation/copy failed (%s:%d)", __FILE__, __LINE__); return -2; } } while (0)

int ft8rx_spectrogram(ft8rx_handle* h, const int16_t* audio, int B, float* grid) {
    if (!h || !audio || !grid || B < 1 || B > h->max_frames) return -1;
    ENTER(h);
    if (need_staging(h)) return -2;
    HIPCHK(h, hipMemcpy(h->d_audio, audio, sizeof(int16_t) * (size_t)B * FT8RX_NSAMP, hipMemcpyHostToDevice));
    ft8rx_ilp_spectrogram(B, h->stream, h->d_audio, h->d_grid, h->T);
    HIPCHK(h, hipStreamSynchronize(h->stream));
    HIPCHK(h, hipMemcpy(grid, h->d_grid, sizeof(float) * (size_t)B * FT8RX_GRID_ROWS * FT8RX_GRID_COLS, hipMemcpyDeviceToHost));
    return 0;
}

int ft8rx_hop_spectrum(ft8rx_handle* h, const int16_t* window3840, float* row) {
    if (!h || !window3840 || !row) return -1;
    ENTER(h);
    if (need_staging(h)) return -2;
    float* d_row = h->d_best_score;                      // any scratch of FT8RX_GRID_COLS floats (NF0MAX >= that): not in use between batches
    HIPCHK(h, hipMemcpyAsync(h->d_audio, window3840, sizeof(int16_t) * 3840, hipMemcpyHostToDevice, h->stream));
    ft8rx_ilp_hop_spectrum(h->stream, h->d_audio, d_row, h->T);
    HIPCHK(h, hipMemcpyAsync(row, d_row, sizeof(float) * FT8RX_GRID_COLS, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    return 0;
}

int ft8rx_sync_search(ft8rx_handle* h, const float* grid, int B, int32_t* f0_idx, int32_t* h0_idx, float* score, int32_t* counts) {
    if (!h || !grid || B < 1 || B > h->max_frames) return -1;
    ENTER(h);
    const ft8rx_config& c = h->cfg;
    HIPCHK(h, hipMemcpy(h->d_grid, grid, sizeof(float) * (size_t)B * FT8RX_GRID_ROWS * FT8RX_GRID_COLS, hipMemcpyHostToDevice));
    const int ntile = (c.f0_hi - c.f0_lo + 15) / 16;
    launch_sync(h->d_grid, h->d_best_score, h->d_best_h0, c, B, h->stream);
    k_topk<<<B, 1024, 0, h->stream>>>(h->d_best_score, h->d_best_h0, h->d_rec, h->d_ncand, c, nullptr, nullptr, nullptr);
    HIPCHK(h, hipStreamSynchronize(h->stream));
    std::vector<ft8rx_record> rec((size_t)B * MAXC);
    HIPCHK(h, hipMemcpy(rec.data(), h->d_rec, sizeof(ft8rx_record) * rec.size(), hipMemcpyDeviceToHost));
    HIPCHK(h, hipMemcpy(counts, h->d_ncand, sizeof(int32_t) * B, hipMemcpyDeviceToHost));
    for (int f = 0; f < B; f++) for (int i = 0; i < c.max_cands; i++) {
        const ft8rx_record& r = rec[(size_t)f * MAXC + i];
        size_t o = (size_t)f * c.max_cands + i;
        f0_idx[o] = r.f0_idx; h0_idx[o] = r.h0_idx; score[o] = r.score;
    }
    // restore the 1.0 row in case the caller's grid differed
    return 0;
}

int ft8rx_sync_scores(ft8rx_handle* h, const float* grid, int B, int f0_lo, int f0_hi, float* score, int32_t* h0_idx) {
    if (!h || !grid || !score || !h0_idx || B < 1 || B > h->max_frames) return -1;
    if (f0_lo < 4 || f0_hi <= f0_lo || f0_hi > FT8RX_GRID_COLS - 15 || f0_hi - f0_lo > NF0MAX) {
        set_err(h, "ft8rx_sync_scores: f0 range [%d, %d) outside [4, %d]", f0_lo, f0_hi, FT8RX_GRID_COLS - 15); return -1; }
    ENTER(h);
    ft8rx_config c = h->cfg;
    c.f0_lo = f0_lo; c.f0_hi = f0_hi;
    HIPCHK(h, hipMemcpy(h->d_grid, grid, sizeof(float) * (size_t)B * FT8RX_GRID_ROWS * FT8RX_GRID_COLS, hipMemcpyHostToDevice));
    const int nf0 = f0_hi - f0_lo, ntile = (nf0 + 15) / 16;
    launch_sync(h->d_grid, h->d_best_score, h->d_best_h0, c, B, h->stream);
    HIPCHK(h, hipGetLastError());
    HIPCHK(h, hipMemcpy2DAsync(score, sizeof(float) * nf0, h->d_best_score, sizeof(float) * NF0MAX, sizeof(float) * nf0, B, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipMemcpy2DAsync(h0_idx, sizeof(int32_t) * nf0, h->d_best_h0, sizeof(int32_t) * NF0MAX, sizeof(int32_t) * nf0, B, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    return 0;
}

int ft8rx_llr_grid(ft8rx_handle* h, const float* grid, int B, int n, const int32_t* frame, const int32_t* f0_idx,
                   const int32_t* h0_idx, float* llr, float* sd, int32_t* snr) {
    if (!h || !grid || B < 1 || B > h->max_frames || n < 1) return -1;
    ENTER(h);
    HIPCHK(h, hipMemcpy(h->d_grid, grid, sizeof(float) * (size_t)B * FT8RX_GRID_ROWS * FT8RX_GRID_COLS, hipMemcpyHostToDevice));
    std::vector<int32_t> trip(3 * (size_t)n);
    for (int i = 0; i < n; i++) { trip[3 * i] = frame[i]; trip[3 * i + 1] = f0_idx[i]; trip[3 * i + 2] = h0_idx[i]; }
    Scratch S{h};
    int32_t* d_trip = S.put(trip.data(), trip.size()); NEED(d_trip);
    float* d_llr = S.get<float>((size_t)n * 174); NEED(d_llr);
    float* d_sd = S.get<float>(n); NEED(d_sd);
    int32_t* d_snr = S.get<int32_t>(n); NEED(d_snr);
    k_grid_llr<<<n, 64, 0, h->stream>>>(h->d_grid, nullptr, nullptr, d_llr, h->cfg, d_trip, d_sd, d_snr, nullptr, nullptr, nullptr, n);
    HIPCHK(h, hipStreamSynchronize(h->stream));
    HIPCHK(h, hipMemcpy(llr, d_llr, sizeof(float) * (size_t)n * 174, hipMemcpyDeviceToHost));
    HIPCHK(h, hipMemcpy(sd, d_sd, sizeof(float) * n, hipMemcpyDeviceToHost));
    HIPCHK(h, hipMemcpy(snr, d_snr, sizeof(int32_t) * n, hipMemcpyDeviceToHost));
    return 0;
}

int ft8rx_cycle_spectrum(ft8rx_handle* h, const int16_t* audio, int B, float* spec) {
    if (!h || !audio || !spec || B < 1 || B > h->max_frames) return -1;
    ENTER(h);
    if (need_staging(h)) return -2;
    HIPCHK(h, hipMemcpy(h->d_audio, audio, sizeof(int16_t) * (size_t)B * FT8RX_NSAMP, hipMemcpyHostToDevice));
    k_cyc_a<<<dim3(40, B), 256, 0, h->stream>>>(h->d_audio, h->d_A, h->T);
    k_cyc_bc<<<dim3(CYC_BC_GRID, B), 256, 0, h->stream>>>(h->d_A, h->d_spec, h->T);
    HIPCHK(h, hipStreamSynchronize(h->stream));
    HIPCHK(h, hipMemcpy(spec, h->d_spec, sizeof(cpx) * (size_t)B * FT8RX_SPEC_BINS, hipMemcpyDeviceToHost));
    return 0;
}

int ft8rx_fine(ft8rx_handle* h, const float* spec, int B, int n, const int32_t* frame, const int32_t* f0_idx, const int32_t* h0_idx,
               int32_t* ret, int32_t* ttweak, int32_t* ftweak, int32_t* nsync, float* llr, float* sd, int32_t* snr, float* sgrid) {
    if (!h || !spec || B < 1 || B > h->max_frames || n < 1) return -1;
    ENTER(h);
    HIPCHK(h, hipMemcpy(h->d_spec, spec, sizeof(cpx) * (size_t)B * FT8RX_SPEC_BINS, hipMemcpyHostToDevice));
    std::vector<int32_t> trip(3 * (size_t)n);
    for (int i = 0; i < n; i++) { trip[3 * i] = frame[i]; trip[3 * i + 1] = f0_idx[i]; trip[3 * i + 2] = h0_idx[i]; }
    Scratch S{h};
    int32_t* d_trip = S.put(trip.data(), trip.size()); NEED(d_trip);
    float* d_llr = S.get<float>((size_t)n * 174); NEED(d_llr);
    HIPCHK(h, hipMemset(d_llr, 0, sizeof(float) * (size_t)n * 174));
    float* d_sd = S.get<float>(n); NEED(d_sd);
    int32_t* d_out = S.get<int32_t>((size_t)n * 5); NEED(d_out);
    float* d_sg = sgrid ? S.get<float>((size_t)n * 632) : nullptr; if (sgrid) NEED(d_sg);
    ft8rx_ilp_fine(n, h->stream, h->d_spec, nullptr, nullptr, d_llr, h->T, h->cfg, d_trip, d_out, d_sd, d_sg, WorkList{nullptr, nullptr});
    k_fine_td<<<n, FINE_NT, 0, h->stream>>>(h->d_spec, nullptr, nullptr, d_llr, h->T, h->cfg, d_trip, d_out, d_sd, d_sg, WorkList{nullptr, nullptr});   // triples k_fine leaves out
    HIPCHK(h, hipStreamSynchronize(h->stream));
    std::vector<int32_t> o((size_t)n * 5);
    HIPCHK(h, hipMemcpy(o.data(), d_out, sizeof(int32_t) * o.size(), hipMemcpyDeviceToHost));
    for (int i = 0; i < n; i++) { ret[i] = o[5 * i]; ttweak[i] = o[5 * i + 1]; ftweak[i] = o[5 * i + 2]; nsync[i] = o[5 * i + 3]; snr[i] = o[5 * i + 4]; }
    HIPCHK(h, hipMemcpy(llr, d_llr, sizeof(float) * (size_t)n * 174, hipMemcpyDeviceToHost));
    HIPCHK(h, hipMemcpy(sd, d_sd, sizeof(float) * n, hipMemcpyDeviceToHost));
    if (sgrid) HIPCHK(h, hipMemcpy(sgrid, d_sg, sizeof(float) * (size_t)n * 632, hipMemcpyDeviceToHost));
    return 0;
}

int ft8rx_ldpc(ft8rx_handle* h, const float* llr, int n, int max_ncheck0, int max_iters, int32_t* ok, uint64_t* msg_lo,
               uint64_t* msg_hi, int32_t* n_its, int32_t* has_out, float* llr_out) {
    if (!h || !llr || n < 1) return -1;
    ENTER(h);
    Scratch S{h};
    float* d_in = S.put(llr, (size_t)n * 174); NEED(d_in);
    float* d_out = S.get<float>((size_t)n * 174); NEED(d_out);
    HIPCHK(h, hipMemset(d_out, 0, sizeof(float) * (size_t)n * 174));
    Att* d_att = S.get<Att>(n); NEED(d_att);
    k_bp<<<n, 64, 0, h->stream>>>(2, d_in, nullptr, nullptr, nullptr, d_att, d_out, nullptr, nullptr, h->cfg, max_ncheck0, max_iters, WorkList{nullptr, nullptr}, 0, 1);
    HIPCHK(h, hipStreamSynchronize(h->stream));
    std::vector<Att> a(n);
    HIPCHK(h, hipMemcpy(a.data(), d_att, sizeof(Att) * n, hipMemcpyDeviceToHost));
    for (int i = 0; i < n; i++) { ok[i] = a[i].ok; msg_lo[i] = a[i].lo; msg_hi[i] = a[i].hi; n_its[i] = a[i].ok ? a[i].n_its : -1; has_out[i] = a[i].has_out; }
    if (llr_out) HIPCHK(h, hipMemcpy(llr_out, d_out, sizeof(float) * (size_t)n * 174, hipMemcpyDeviceToHost));
    return 0;
}

int ft8rx_osd_ext(ft8rx_handle* h, const float* llr, int n, int singleflips, int doubleflips, int tripleflips, int max_hd,
                  int32_t* ok, uint64_t* msg_lo, uint64_t* msg_hi, int32_t* trial, int32_t* hd) {
    if (!h || !llr || n < 1) return -1;
    if (singleflips < 0 || singleflips > OSD_MAXFLIP || doubleflips < 0 || doubleflips > OSD_MAXFLIP || tripleflips < 0 || tripleflips > 40 ||
        max_hd < 0 || max_hd > 174) { set_err(h, "ft8rx_osd: flip counts out of range"); return -1; }
    const std::vector<uint32_t> tr = osd_trial_table(singleflips, doubleflips, tripleflips);
    if (tr.size() > OSD_MAXTRIALS) { set_err(h, "ft8rx_osd: %zu trials exceed %d", tr.size(), OSD_MAXTRIALS); return -1; }
    ENTER(h);
    Scratch S{h};
    float* d_in = S.put(llr, (size_t)n * 174); NEED(d_in);
    uint32_t* d_tr = S.put(tr.data(), tr.size()); NEED(d_tr);
    Att* d_att = S.get<Att>(n); NEED(d_att);
    int32_t* d_nan = S.get<int32_t>((size_t)n + 1); NEED(d_nan);            // [0] = length of the list of NaN vectors, then the list
    HIPCHK(h, hipMemsetAsync(d_nan, 0, sizeof(int32_t), h->stream));
    const WorkList nanl{d_nan + 1, d_nan};
    const bool osd_wide = osd_nflip(singleflips, tripleflips) > OSD_FLIPS_A;
    (osd_wide ? k_osd_wide : k_osd)<<<n, 64, 0, h->stream>>>(
        2, d_in, nullptr, nullptr, nullptr, nullptr, d_att, nullptr, nullptr, d_tr, (int)tr.size(), osd_nflip(singleflips, tripleflips), max_hd,
        WorkList{nullptr, nullptr}, nanl);
    (osd_wide ? k_osd_nan_wide : k_osd_nan)<<<OSD_NAN_GRID, 64, 0, h->stream>>>(
        2, d_in, nullptr, nullptr, nullptr, nullptr, d_att, nullptr, nullptr, d_tr, (int)tr.size(), osd_nflip(singleflips, tripleflips), max_hd, nanl);
    HIPCHK(h, hipStreamSynchronize(h->stream));
    std::vector<Att> a(n);
    HIPCHK(h, hipMemcpy(a.data(), d_att, sizeof(Att) * n, hipMemcpyDeviceToHost));
    for (int i = 0; i < n; i++) {
        ok[i] = a[i].ok; msg_lo[i] = a[i].lo; msg_hi[i] = a[i].hi; trial[i] = a[i].ok ? a[i].n_its : -1;
        if (hd) hd[i] = a[i].ok ? a[i].pad[0] : -1;
    }
    return 0;
}

int ft8rx_osd(ft8rx_handle* h, const float* llr, int n, int singleflips, int doubleflips, int32_t* ok, uint64_t* msg_lo,
              uint64_t* msg_hi, int32_t* trial) {
    return ft8rx_osd_ext(h, llr, n, singleflips, doubleflips, 0, 0, ok, msg_lo, msg_hi, trial, nullptr);
}

int ft8rx_crc_valid(ft8rx_handle* h, const float* cw91, int n, int32_t* res, uint64_t* msg_lo, uint64_t* msg_hi) {
    if (!h || !cw91 || n < 1) return -1;
    ENTER(h);
    Scratch S{h};
    float* d_in = S.put(cw91, (size_t)n * 91); NEED(d_in);
    int32_t* d_res = S.get<int32_t>(n); NEED(d_res);
    uint64_t* d_lo = S.get<uint64_t>(n); NEED(d_lo);
    uint64_t* d_hi = S.get<uint64_t>(n); NEED(d_hi);
    k_crc_probe<<<n, 64, 0, h->stream>>>(d_in, n, d_res, d_lo, d_hi);
    HIPCHK(h, hipStreamSynchronize(h->stream));
    HIPCHK(h, hipMemcpy(res, d_res, sizeof(int32_t) * n, hipMemcpyDeviceToHost));
    HIPCHK(h, hipMemcpy(msg_lo, d_lo, sizeof(uint64_t) * n, hipMemcpyDeviceToHost));
    HIPCHK(h, hipMemcpy(msg_hi, d_hi, sizeof(uint64_t) * n, hipMemcpyDeviceToHost));
    return 0;
}

int ft8rx_valid77(ft8rx_handle* h, const uint64_t* msg_lo, const uint64_t* msg_hi, int n, int32_t* valid) {
    if (!h || !msg_lo || !msg_hi || n < 1) return -1;
    ENTER(h);
    Scratch S{h};
    uint64_t* d_lo = S.put(msg_lo, n); NEED(d_lo);
    uint64_t* d_hi = S.put(msg_hi, n); NEED(d_hi);
    int32_t* d_v = S.get<int32_t>(n); NEED(d_v);
    k_valid_probe<<<(n + 255) / 256, 256, 0, h->stream>>>(d_lo, d_hi, n, d_v);
    HIPCHK(h, hipStreamSynchronize(h->stream));
    HIPCHK(h, hipMemcpy(valid, d_v, sizeof(int32_t) * n, hipMemcpyDeviceToHost));
    return 0;
}

int16_t* ft8rx_staging_audio(ft8rx_handle* h) {
    if (!h || hipSetDevice(h->device) != hipSuccess || need_staging(h)) return nullptr;
    return h->d_audio;
}

// Asynchronous D2H copies for a consumer of device-side results (the gather on rank `dst`, pyft8_amd/distributed.py), on the handle's
// RESULT-COPY stream: the one stream of the handle no decode kernel ever waits behind.  The runtime maps streams onto four hardware
// queues and the commands of a queue execute in order, so a copy issued on any OTHER stream of the process (a torch side stream)
// lands in a queue it shares with one of the decode streams and holds that stream's kernels back for as long as it runs -- 12 MB per
// step (rank 0's load in an 8-GPU config-1 job) cost 4.7 % of the step that way (profiles/r06_notes.md).
int ft8rx_d2h_async(ft8rx_handle* h, void* dst, const void* d_src, uint64_t bytes, int32_t* ticket) {
    if (!h || !dst || !d_src || !ticket) return -1;
    HIPCHK(h, hipSetDevice(h->device));
    const uint32_t id = h->d2h_next;
    hipEvent_t& e = h->d2h_ev[id & 31];
    if (!e) HIPCHK(h, hipEventCreateWithFlags(&e, hipEventDisableTiming));
    else if (id >= 32 && hipEventQuery(e) == hipErrorNotReady) HIPCHK(h, hipEventSynchronize(e));      // 32 copies in flight: wait for the oldest
    if (bytes) HIPCHK(h, hipMemcpyAsync(dst, d_src, (size_t)bytes, hipMemcpyDeviceToHost, h->copy_s));
    HIPCHK(h, hipEventRecord(e, h->copy_s));
    h->d2h_next = id + 1;
    *ticket = (int32_t)(id & 0x7fffffffu);
    return 0;
}
int ft8rx_d2h_query(ft8rx_handle* h, int32_t ticket) {              // 1 = the copy has landed, 0 = not yet, < 0 = error
    if (!h || ticket < 0) return -1;
    const uint32_t id = (uint32_t)ticket, next = h->d2h_next & 0x7fffffffu;
    const uint32_t age = (next - id) & 0x7fffffffu;
    if (age == 0u || age > 0x40000000u) { set_err(h, "ft8rx_d2h_query: ticket %d was never issued", ticket); return -1; }
    if (age > 32u) return 1;                                        // its event has been reused since: long done
    const hipError_t r = hipEventQuery(h->d2h_ev[id & 31]);
    if (r == hipSuccess) return 1;
    if (r == hipErrorNotReady) return 0;
    set_err(h, "ft8rx_d2h_query: %s", hipGetErrorString(r));
    return -2;
}
void* ft8rx_d2h_event(ft8rx_handle* h, int32_t ticket) {            // the HIP event behind that copy (for ft8rx_packed_output_fence), or NULL
    if (!h || ticket < 0) return nullptr;
    const uint32_t id = (uint32_t)ticket, next = h->d2h_next & 0x7fffffffu;
    const uint32_t age = (next - id) & 0x7fffffffu;
    return (age >= 1u && age <= 32u) ? (void*)h->d2h_ev[id & 31] : nullptr;
}

int ft8rx_copy_to_host(ft8rx_handle* h, void* dst, const void* d_src, uint64_t bytes) {
    if (!h || !dst || !d_src) return -1;
    ENTER(h);
    HIPCHK(h, hipMemcpy(dst, d_src, (size_t)bytes, hipMemcpyDeviceToHost));
    return 0;
}

int ft8rx_subtract(ft8rx_handle* h, int16_t* d_audio, int B, ft8rx_subsig* sigs, const int32_t* counts, int max_sigs,
                   int refine, float* audio_f32_out) {
    if (!h || !d_audio || !sigs || !counts) return -1;
    if (B < 1 || B > h->max_frames || max_sigs < 1 || max_sigs > 256 || refine < 0 || refine > 3) { set_err(h, "ft8rx_subtract: bad n_frames / max_sigs / refine"); return -1; }
    ENTER(h);
    HIPCHK(h, hipStreamSynchronize(h->stream));
    if (!h->d_wf) {              // first use: working buffers for max_frames frames and the GFSK pulse tables (transmitter.py:41-50)
        const size_t MB = (size_t)h->max_frames;
        int rc = dalloc(h, &h->d_wf, MB * FT8RX_NSAMP);
        rc |= dalloc(h, &h->d_part, MB * SUB_NCH * 20);           // also holds the refinement scan: [SUB_MAXSHIFT][SUB_NCH] <= [SUB_NCH][20]
        rc |= dalloc(h, &h->d_pulse, (size_t)5760);
        rc |= dalloc(h, &h->d_pc, (size_t)5760);
        rc |= dalloc(h, &h->d_sigcnt, MB);
        if (rc) return -2;
        std::vector<double> pulse(5760), pc(5760);
        const double c = M_PI * sqrt(2.0 / log(2.0)), bt = 2.0;
        double acc = 0.0;
        for (int i = 0; i < 5760; i++) {
            const double tt = ((double)i - 1.5 * 1920.0) / 1920.0;
            pulse[i] = 0.5 * (erf(c * bt * (tt + 0.5)) - erf(c * bt * (tt - 0.5)));
            acc += pulse[i]; pc[i] = acc;
        }
        HIPCHK(h, hipMemcpy(h->d_pulse, pulse.data(), sizeof(double) * 5760, hipMemcpyHostToDevice));
        HIPCHK(h, hipMemcpy(h->d_pc, pc.data(), sizeof(double) * 5760, hipMemcpyHostToDevice));
    }
    if (h->sig_cap < max_sigs) {
        ft8rx_subsig* p = nullptr;
        if (dalloc(h, &p, (size_t)h->max_frames * max_sigs)) return -2;       // the smaller one stays in the handle's allocation list
        h->d_sigs = p; h->sig_cap = max_sigs;
    }
    int nmax = 0;
    for (int f = 0; f < B; f++) {
        if (counts[f] < 0 || counts[f] > max_sigs) { set_err(h, "ft8rx_subtract: counts[%d] = %d outside [0, %d]", f, counts[f], max_sigs); return -1; }
        if (counts[f] > nmax) nmax = counts[f];
    }
    HIPCHK(h, hipMemcpyAsync(h->d_sigs, sigs, sizeof(ft8rx_subsig) * (size_t)B * max_sigs, hipMemcpyHostToDevice, h->stream));
    HIPCHK(h, hipMemcpyAsync(h->d_sigcnt, counts, sizeof(int32_t) * B, hipMemcpyHostToDevice, h->stream));
    const size_t n = (size_t)B * FT8RX_NSAMP;
    k_sub_to_f32<<<(unsigned)((n + 255) / 256), 256, 0, h->stream>>>(d_audio, h->d_wf, n);
    const SubTables T{h->d_pulse, h->d_pc};
    // refinement scans: coarse (10 ms / 0.25 Hz around the decoder's origin, which by the search-grid conventions sits ~75 ms late and
    // ~1.9 Hz low), then fine (2.5 ms / 0.0625 Hz)
    SubShifts coarse, fine;
    coarse.stride = 1; fine.stride = 1;          // decimating the scan (tried 8 / 2) aliases neighbouring signals into the sum: 1 of 233 origins locked 70 ms off
    coarse.n = 16; for (int i = 0; i < 16; i++) coarse.shift[i] = -1680 + 120 * i;          // -140 .. +10 ms
    fine.n = 9;    for (int i = 0; i < 9; i++) fine.shift[i] = -120 + 30 * i;               // -10 .. +10 ms
    for (int i = 9; i < SUB_MAXSHIFT; i++) fine.shift[i] = 0;
    if (refine == 2 && !h->d_zdec) {
        const size_t MB = (size_t)h->max_frames;
        int rc = dalloc(h, &h->d_zdec, MB * SUBD_NZ);
        rc |= dalloc(h, &h->d_model, MB * SUBD_N);
        rc |= dalloc(h, &h->d_adec, MB * (SUBD_N + 1));
        rc |= dalloc(h, &h->d_subctx, MB);
        if (rc) return -2;
    }
    // refine = 2: the same steps on a decimated baseband copy (kernels/subtract.hpp): time shifts in units of 32 samples
    SubShifts dcoarse, dfine;
    dcoarse.stride = 1; dfine.stride = 1;
    dcoarse.n = 15; for (int i = 0; i < 15; i++) dcoarse.shift[i] = SUBD_D * (-52 + 4 * i);        // -138.7 .. +10.7 ms in 10.7 ms steps
    dcoarse.shift[15] = 0;
    dfine.n = 9;    for (int i = 0; i < 9; i++) dfine.shift[i] = SUBD_D * (-4 + i);                // -10.7 .. +10.7 ms in 2.67 ms steps
    for (int i = 9; i < SUB_MAXSHIFT; i++) dfine.shift[i] = 0;
    const dim3 gmodel((SUBD_N + 255) / 256, B);
    Tables Tones = h->T;                                     // refine = 3: the experiment's slices have no edge tapers
    if (refine == 3) {
        if (!h->d_ones) {
            if (dalloc(h, &h->d_ones, (size_t)100)) return -2;
            double ones[100]; for (int i = 0; i < 100; i++) ones[i] = 1.0;
            HIPCHK(h, hipMemcpy(h->d_ones, ones, sizeof(ones), hipMemcpyHostToDevice));
        }
        Tones.taper = h->d_ones;
    }
    for (int s = 0; s < nmax; s++) {
        if (refine == 3) {
            // Candidate.refine_time_origin (receiver_sub.py:58-72) on the spectrum of the residual so far, then subtract_signal as is
            k_cyc_a_f32<<<dim3(40, B), 256, 0, h->stream>>>(h->d_wf, h->d_A, h->T);
            k_cyc_bc<<<dim3(CYC_BC_GRID, B), 256, 0, h->stream>>>(h->d_A, h->d_spec, h->T);
            k_refine3<<<B, FINE_NT, 0, h->stream>>>(h->d_spec, h->d_sigs, h->d_sigcnt, max_sigs, s, Tones);
            k_sub_accum<<<dim3(SUB_GRIDX, B), 256, 0, h->stream>>>(h->d_wf, h->d_sigs, h->d_sigcnt, max_sigs, s, T, h->d_part);
            k_sub_apply<<<dim3(SUB_GRIDX, B), 256, 0, h->stream>>>(h->d_wf, h->d_sigs, h->d_sigcnt, max_sigs, s, T, h->d_part);
            continue;
        }
        if (refine == 2) {
            k_subd_mix<<<dim3(SUBD_NZ / 256, B), 256, 0, h->stream>>>(h->d_wf, h->d_sigs, h->d_sigcnt, max_sigs, s, h->d_zdec, h->d_subctx);
            k_subd_model<<<gmodel, 256, 0, h->stream>>>(h->d_sigs, h->d_sigcnt, max_sigs, s, T, h->d_subctx, h->d_model);
            k_subd_scan<<<B, 256, 0, h->stream>>>(h->d_zdec, h->d_model, h->d_sigs, h->d_sigcnt, max_sigs, s, h->d_subctx, dcoarse, h->d_part);
            k_sub_pick<<<B, 256, 0, h->stream>>>(h->d_sigs, h->d_sigcnt, max_sigs, s, dcoarse, h->d_part, -1.0f, 0.0625f, 113, SUBD_D * SUBD_CH);
            k_subd_model<<<gmodel, 256, 0, h->stream>>>(h->d_sigs, h->d_sigcnt, max_sigs, s, T, h->d_subctx, h->d_model);
            k_subd_scan<<<B, 256, 0, h->stream>>>(h->d_zdec, h->d_model, h->d_sigs, h->d_sigcnt, max_sigs, s, h->d_subctx, dfine, h->d_part);
            k_sub_pick<<<B, 256, 0, h->stream>>>(h->d_sigs, h->d_sigcnt, max_sigs, s, dfine, h->d_part, 0.4375f, 0.015625f, 9, SUBD_D * SUBD_CH);
            k_subd_model<<<gmodel, 256, 0, h->stream>>>(h->d_sigs, h->d_sigcnt, max_sigs, s, T, h->d_subctx, h->d_model);
            k_subd_accum<<<B, 256, 0, h->stream>>>(h->d_zdec, h->d_model, h->d_sigs, h->d_sigcnt, max_sigs, s, h->d_subctx, h->d_adec);
            k_subd_apply<<<dim3(SUB_GRIDX, B), 256, 0, h->stream>>>(h->d_wf, h->d_sigs, h->d_sigcnt, max_sigs, s, T, h->d_adec);
            continue;
        }
        if (refine) {
            k_sub_scan<<<dim3(SUB_GRIDX, B), 256, 0, h->stream>>>(h->d_wf, h->d_sigs, h->d_sigcnt, max_sigs, s, T, coarse, h->d_part);
            k_sub_pick<<<B, 256, 0, h->stream>>>(h->d_sigs, h->d_sigcnt, max_sigs, s, coarse, h->d_part, -1.0f, 0.0625f, 113, SUB_CH);   // signal - model: -1 .. +6 Hz; the sum is coherent over 12.6 s, so the grid must be as fine as 1/16 Hz
            k_sub_scan<<<dim3(SUB_GRIDX, B), 256, 0, h->stream>>>(h->d_wf, h->d_sigs, h->d_sigcnt, max_sigs, s, T, fine, h->d_part);
            k_sub_pick<<<B, 256, 0, h->stream>>>(h->d_sigs, h->d_sigcnt, max_sigs, s, fine, h->d_part, 0.4375f, 0.015625f, 9, SUB_CH);   // around the +0.5 Hz the coarse step left
        }
        k_sub_accum<<<dim3(SUB_GRIDX, B), 256, 0, h->stream>>>(h->d_wf, h->d_sigs, h->d_sigcnt, max_sigs, s, T, h->d_part);
        k_sub_apply<<<dim3(SUB_GRIDX, B), 256, 0, h->stream>>>(h->d_wf, h->d_sigs, h->d_sigcnt, max_sigs, s, T, h->d_part);
    }
    k_sub_to_i16<<<(unsigned)((n + 255) / 256), 256, 0, h->stream>>>(h->d_wf, d_audio, n);
    HIPCHK(h, hipGetLastError());
    if (audio_f32_out) HIPCHK(h, hipMemcpyAsync(audio_f32_out, h->d_wf, sizeof(float) * n, hipMemcpyDeviceToHost, h->stream));
    if (refine) HIPCHK(h, hipMemcpyAsync(sigs, h->d_sigs, sizeof(ft8rx_subsig) * (size_t)B * max_sigs, hipMemcpyDeviceToHost, h->stream));   // the refined origins
    HIPCHK(h, hipStreamSynchronize(h->stream));
    return 0;
}

int ft8rx_encode_tones(const uint64_t* msg_lo, const uint64_t* msg_hi, int n, uint8_t* tones) {
    if (!msg_lo || !msg_hi || !tones || n < 0) return -1;
    for (int i = 0; i < n; i++) hostmsg::encode_tones(msg_lo[i], msg_hi[i], tones + (size_t)i * 79);
    return 0;
}

void* ft8rx_alloc_host(ft8rx_handle* h, uint64_t bytes) {
    if (!h || bytes == 0) return nullptr;
    void* p = nullptr;
    if (hipSetDevice(h->device) != hipSuccess || hipHostMalloc(&p, (size_t)bytes, hipHostMallocDefault) != hipSuccess) {
        set_err(h, "ft8rx_alloc_host: hipHostMalloc(%llu) failed", (unsigned long long)bytes);
        return nullptr;
    }
    return p;
}

int ft8rx_free_host(ft8rx_handle* h, void* p) {          // h may be NULL (buffers can outlive the handle that allocated them)
    if (!p) return -1;
    const hipError_t e = hipHostFree(p);
    if (e != hipSuccess) { if (h) set_err(h, "hipHostFree failed: %s", hipGetErrorString(e)); return -2; }
    return 0;
}

int ft8rx_synth_frames_ex(ft8rx_handle* h, uint64_t seed, int first_index, int n_frames, int n_signals,
                          const void* signal_table, int signal_bytes, const double* pulse_cumsum, int16_t* d_audio, int no_noise) {
    if (!h || !signal_table || !pulse_cumsum || !d_audio || n_frames < 1 || n_signals < 0 || n_signals > 64) return -1;
    if (signal_bytes != (int)sizeof(SynthSig)) { set_err(h, "ft8rx_synth_frames: signal record is %d bytes, expected %d", signal_bytes, (int)sizeof(SynthSig)); return -1; }
    ENTER(h);
    Scratch S{h};
    SynthSig* d_s = S.put((const SynthSig*)signal_table, (size_t)n_frames * (n_signals ? n_signals : 1)); NEED(d_s);
    double* d_q = S.put(pulse_cumsum, 5761); NEED(d_q);
    k_synth<<<dim3((FT8RX_NSAMP / 4 + 255) / 256, n_frames), 256, 0, h->stream>>>(d_audio, d_s, n_signals, d_q, (uint32_t)seed, (uint32_t)(seed >> 32), first_index, no_noise);
    HIPCHK(h, hipStreamSynchronize(h->stream));
    return 0;
}

int ft8rx_synth_frames(ft8rx_handle* h, uint64_t seed, int first_index, int n_frames, int n_signals,
                       const void* signal_table, int signal_bytes, const double* pulse_cumsum, int16_t* d_audio) {
    return ft8rx_synth_frames_ex(h, seed, first_index, n_frames, n_signals, signal_table, signal_bytes, pulse_cumsum, d_audio, 0);
}

int ft8rx_package_batch(const ft8rx_record* records, const int32_t* counts, const ft8rx_event* events, const int32_t* event_counts,
                        int n_frames, int max_cands, ft8rx_message* out, int max_msgs, int32_t* out_counts, int n_threads,
                        ft8rx_hashes* table, int32_t* flags) {
    return hostmsg::package_batch(records, counts, events, event_counts, n_frames, max_cands, out, max_msgs, out_counts, n_threads,
                                  table ? &table->H : nullptr, flags);
}

int ft8rx_package_packed(const void* packed, uint64_t bytes, int frame_lo, int n_frames, ft8rx_message* out, int max_msgs,
                         int32_t* out_counts, int n_threads, ft8rx_hashes* table, int32_t* flags) {
    return hostmsg::package_packed(packed, bytes, frame_lo, n_frames, out, max_msgs, out_counts, n_threads, table ? &table->H : nullptr, flags);
}

int ft8rx_merge_messages(ft8rx_message* out, int32_t* out_counts, int max_out, const ft8rx_message* add, const int32_t* add_counts,
                         int max_add, int n_frames, int pass_tag, int drop_osd, ft8rx_message* fresh, int32_t* fresh_counts) {
    if (!out || !out_counts || !add || !add_counts || n_frames < 0 || max_out < 1 || max_add < 1) return -1;
    hostmsg::merge_messages(out, out_counts, max_out, add, add_counts, max_add, n_frames, pass_tag, drop_osd, fresh, fresh_counts);
    return 0;
}

int ft8rx_subtraction_list(const ft8rx_message* msgs, const int32_t* counts, int max_msgs, const ft8rx_record* records, int max_cands,
                           int n_frames, int min_snr, ft8rx_subsig* sigs, int max_sigs, int32_t* sig_counts) {
    if (!msgs || !counts || !records || !sigs || !sig_counts || n_frames < 0 || max_msgs < 1 || max_cands < 1 || max_sigs < 1) return -1;
    return hostmsg::subtraction_list(msgs, counts, max_msgs, records, max_cands, n_frames, min_snr, sigs, max_sigs, sig_counts);
}

ft8rx_hashes* ft8rx_hashes_create(void) { return new (std::nothrow) ft8rx_hashes(); }
void ft8rx_hashes_destroy(ft8rx_hashes* t) { delete t; }
int ft8rx_hashes_clear(ft8rx_hashes* t) { if (!t) return -1; t->H.clear(); return 0; }
int ft8rx_hashes_add(ft8rx_hashes* t, const char* call) { if (!t || !call) return -1; t->H.add(call); return 0; }
int ft8rx_hashes_size(const ft8rx_hashes* t) { return t ? (int)t->H.size() : -1; }

int ft8rx_set_reject_log(const char* path) { hostmsg::set_reject_log(path); return 0; }

#ifdef FINE_TIMING
int ft8rx_debug_fine_times(ft8rx_handle* h, unsigned long long* out32, int reset) {      // timing-only builds (tools/fine_timing.sh)
    if (!h) return -1;
    ENTER(h);
    return ft8rx_ilp_fine_times(out32, reset);          // the instrumented kernel lives in the second translation unit
}
#endif

#ifdef BP_TIMING
int ft8rx_debug_bp_times(ft8rx_handle* h, unsigned long long* out8, int reset) {         // timing-only builds (tools/bp_timing.py)
    if (!h) return -1;
    ENTER(h);
    if (hipDeviceSynchronize() != hipSuccess) return -2;
    std::vector<unsigned long long> all((size_t)65536 * 8);
    if (out8) {
        if (hipMemcpyFromSymbol(all.data(), HIP_SYMBOL(g_bp_t), sizeof(unsigned long long) * all.size()) != hipSuccess) return -2;
        for (int i = 0; i < 8; i++) out8[i] = 0;
        for (size_t b = 0; b < 65536; b++) for (int i = 0; i < 8; i++) out8[i] += all[b * 8 + i];
    }
    if (reset) { std::fill(all.begin(), all.end(), 0ull); if (hipMemcpyToSymbol(HIP_SYMBOL(g_bp_t), all.data(), sizeof(unsigned long long) * all.size()) != hipSuccess) return -2; }
    return 0;
}
#endif
#ifdef OSD_TIMING
int ft8rx_debug_osd_times(ft8rx_handle* h, unsigned long long* out16, int reset) {       // timing-only builds (tools/osd_timing.py)
    if (!h) return -1;
    ENTER(h);
    if (hipDeviceSynchronize() != hipSuccess) return -2;
    std::vector<unsigned long long> all((size_t)32768 * 10);
    if (out16) {
        if (hipMemcpyFromSymbol(all.data(), HIP_SYMBOL(g_osd_t), sizeof(unsigned long long) * all.size()) != hipSuccess) return -2;
        for (int i = 0; i < 16; i++) out16[i] = 0;
        for (size_t b = 0; b < 32768; b++) for (int i = 0; i < 10; i++) out16[i == 9 ? 15 : i] += all[b * 10 + i];
    }
    if (reset) { std::fill(all.begin(), all.end(), 0ull); if (hipMemcpyToSymbol(HIP_SYMBOL(g_osd_t), all.data(), sizeof(unsigned long long) * all.size()) != hipSuccess) return -2; }
    return 0;
}
#endif

int ft8rx_math_probe(ft8rx_handle* h, int which, const float* x, int n, float* y) {
    if (!h || !x || !y || n < 1) return -1;
    ENTER(h);
    Scratch S{h};
    if (which == 0 || which == 1) {
        float* d_x = S.put(x, n); NEED(d_x);
        float* d_y = S.get<float>(n); NEED(d_y);
        k_math_probe<<<(n + 255) / 256, 256, 0, h->stream>>>(which, d_x, d_y, n);
        HIPCHK(h, hipStreamSynchronize(h->stream));
        HIPCHK(h, hipMemcpy(y, d_y, sizeof(float) * n, hipMemcpyDeviceToHost));
        return 0;
    }
    if (which == 2) {
        cpx* d_x = S.put((const cpx*)x, n); NEED(d_x);
        cpx* d_y = S.get<cpx>(n); NEED(d_y);
        const size_t lds = 2 * (size_t)n * sizeof(cpx);
        if (n == 1920) k_fft_probe<1920, 8, 4, 4, 5, 3><<<1, 256, lds, h->stream>>>(d_x, d_y, h->T.W1920);
        else if (n == 3200) k_fft_probe<3200, 8, 4, 4, 5, 5><<<1, 256, lds, h->stream>>>(d_x, d_y, h->T.W3200);
        else if (n == 300) k_fft_probe<300, 5, 5, 4, 3><<<1, 256, lds, h->stream>>>(d_x, d_y, h->T.W300);
        else if (n == 320) k_fft_probe<320, 8, 8, 5><<<1, 256, lds, h->stream>>>(d_x, d_y, h->T.W320);
        else { set_err(h, "ft8rx_math_probe: no FFT plan of length %d", n); return -1; }
        HIPCHK(h, hipStreamSynchronize(h->stream));
        HIPCHK(h, hipMemcpy(y, d_y, sizeof(cpx) * n, hipMemcpyDeviceToHost));
        return 0;
    }
    return -1;
}

}  // extern "C"
