// common.hpp -- device tables, attempt records, event log shared by all kernels
// Part of libft8rx.so.  Included by both translation units: ft8rx.hip (everything) and ft8rx_ilp.hip (FT8RX_ILP_UNIT: the FFT kernels,
// which only need the first part -- the statically initialised constants and the plain structs).  Tables that ft8rx_create fills at
// run time, and every kernel of this file, exist in the main unit only.
#ifndef FT8RX_COMMON_HPP
#define FT8RX_COMMON_HPP

// ------------------------------------------------------------------------------------ device tables
struct Tables {
    const float* win;        // [3840] Hann (np.hanning) as f32
    const cpx* W1920;        // twiddles
    const cpx* WR3840;       // [FT8RX_GRID_COLS] real-split twiddles e^{-2 pi i k/3840}
    const cpx* W3200;
    const cpx* W96000;
    const cpx* W300;
    const cpx* W320;
    const cpx* WR192k;       // [FT8RX_SPEC_BINS]
    const cpx* W32;
    const double* taper;     // [100]
    const float* K32;        // [1800] K(m), m = -900 .. 899: the REAL factor of the 32-sample Dirichlet kernel, sin(pi (m mod 100) / 100) / sin(pi m / 3200) (fine_fscore)
    const cpx* CS100;        // [6][26] (cos, sin)(2 pi p s / 100), s = 1 .. 6, p = 0 .. 25
    const cpx* TW100;        // [100] e^{-2 pi i m / 100}: twiddles of the 10 x 10 residue transform of the final grid
    const cpx* G1000;        // [1000] e^{i pi (31 r - 100 j) / 3200} for the slice bin k = r + 100 j = -150 .. 849
};

__device__ __constant__ int d_COSTAS[7] = {3, 1, 4, 0, 6, 5, 2};
__device__ __constant__ uint8_t d_PAYSYM[58] = {7,8,9,10,11,12,13,14,15,16,17,18,19,20,21,22,23,24,25,26,27,28,29,30,31,32,33,34,35,
    43,44,45,46,47,48,49,50,51,52,53,54,55,56,57,58,59,60,61,62,63,64,65,66,67,68,69,70,71};
// Work lists of the decode ladder.  Each ladder kernel only has work for the candidates that are still ACTIVE (39 % of the 256 slots
// per frame for the second BP, 27 % for OSD, a few per cent in sparse low-SNR frames), so instead of one mostly-empty block per slot
// a thread-per-candidate kernel (k_worklist, k_select0, k_select1) appends the candidates that go on to a compact list and the
// consumer indexes list x attempts: k_fine and k_osd with a bounded grid whose blocks stride over the items, k_bp with one
// attempt per block (its attempts are short and very uneven).  Entries are chunk-relative candidate ids (frame * MAXC + ci); the
// order is whatever the atomics give -- every attempt writes its own result slot and the host sorts the event log, so results do
// not depend on it.
struct WorkList { int32_t* items; int32_t* count; };

// XCD-aware block map for kernels whose blocks of one FRAME read the same memory (the frame's dB grid): consecutive workgroup ids go
// round-robin over the 8 XCDs, each with an L2 of its own, so with the plain map (frame = id / per) the `per` blocks of a frame land on
// all eight and every XCD fetches the frame's lines again.  Here workgroup id L -> XCD L % 8 takes frame 8 (j / per) + L % 8, block
// j % per (j = L / 8): a frame's blocks share one L2.  Launch ((B + 7) / 8) * 8 * per blocks; false = no such frame (B % 8 != 0).
FT8_DEV bool xcd_frame_map(int L, int per, int B, int& frame, int& idx) {
    const int x = L & 7, j = L >> 3;
    frame = 8 * (j / per) + x; idx = j % per;
    return frame < B;
}
#define XCD_GRID(B, per) ((((B) + 7) / 8) * 8 * (per))
enum { WL_BP0 = 0, WL_FINE = 1, WL_BP1 = 2, WL_BP1B = 3, WL_BP1C = 4, WL_OSD = 5, WL_OSDNAN = 6, WL_N = 7 };      // WL_BP0 lists attempts (x 5), WL_OSDNAN OSD attempts (x 10)

#define W6 ((double)(-0.16666667163372040f))     /* np.float32(-1/6), receiver.py:323,198 */

#ifndef FT8RX_ILP_UNIT                           /* ---- main translation unit only from here on ---- */
// AP masks (reference receiver.py:21-27), copied verbatim as data
__device__ __constant__ int8_t d_AP_CQ[29]   = {0,0,0,0,0, 0,0,0,0,0, 0,0,0,0,0, 0,0,0,0,0, 0,0,0,0,0, 0,1,0,0};
__device__ __constant__ int8_t d_AP_END[3][19] = {{0,1, 1,1,1,1,1, 0,0,1,1,1, 0,1,0,1,0, 0,1},
                                                  {0,1, 1,1,1,1,1, 0,1,0,0,1, 0,1,0,0,0, 0,1},
                                                  {0,1, 1,1,1,1,1, 0,1,0,0,1, 0,0,1,0,0, 0,1}};
// LDPC tables in device memory (copies of ft8_tables.h)
__device__ uint8_t  d_CHK_N[83];
__device__ int16_t  d_CHK_V[83][7];
__device__ uint8_t  d_EDGE_V[522];
__device__ uint8_t  d_EDGE_C[522];
__device__ uint16_t d_VAR_E[174][3];
__device__ uint64_t d_G0[91][3];
__device__ uint64_t d_CHK_MASK[128][3];   // membership mask of check c over the 174 variables (rows >= 83 are zero)

struct Att {               // one decode attempt's outcome
    uint64_t lo, hi;
    int16_t n_its;
    uint8_t ok;            // 1 = accepted
    uint8_t method;        // FT8RX_M_*
    uint8_t nc0;           // initial unsatisfied-check count (BP)
    uint8_t has_out;       // BP left a 174-vector behind (the reference's third return value)
    uint8_t pad[2];
};

FT8_DEV void work_push(const WorkList& w, int cand) { if (w.items) w.items[atomicAdd(w.count, 1)] = cand; }
// Block-aggregated push for thread-per-candidate kernels (256-thread blocks): ONE atomic per block reserves a range -- tens of
// thousands of atomics on a single counter serialise in L2 (measured: k_grid_llr 0.09 -> 0.61 ms with one atomic per candidate).
// Every thread of the block must call this (it contains block barriers).
FT8_DEV void work_push_block(const WorkList& w, bool want, int cand) {
    __shared__ int s_wave[4], s_base;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const uint64_t m = __ballot(want);
    if (lane == 0) s_wave[wv] = __popcll(m);
    __syncthreads();
    if (threadIdx.x == 0) {
        const int tot = s_wave[0] + s_wave[1] + s_wave[2] + s_wave[3];
        s_base = tot ? atomicAdd(w.count, tot) : 0;
    }
    __syncthreads();
    if (want) {
        int off = s_base + __popcll(m & ((1ull << lane) - 1));
        for (int i = 0; i < wv; i++) off += s_wave[i];
        w.items[off] = cand;
    }
}
// grid row accessor with the reference's modulo-750 wrap (receiver.py:240,347,360)
// (branch-free: an always-valid clamped load, then an integer mask selects the grid's initial 1.0 -- a branch or select around
// the load would serialise the loads of a gather loop, one memory round trip per basic block)
FT8_DEV float grid_at(const float* __restrict__ g, int row, int col) {
    row %= 750; if (row < 0) row += 750;
    const bool in = row >= 1 && row <= 375;
    const uint32_t raw = __float_as_uint(g[(in ? row : 1) * FT8RX_GRID_COLS + col]);
    const uint32_t m = in ? 0xFFFFFFFFu : 0u;
    return __uint_as_float((raw & m) | (0x3F800000u & ~m));
}

FT8_DEV void log_event(ft8rx_event* ev, int32_t* evcount, int frame, int cand, int ipass, int slot, int seq,
                       uint64_t lo, uint64_t hi, int valid) {
    if (!ev) return;
    int idx = atomicAdd(&evcount[frame], 1);
    if (idx < FT8RX_EVENT_CAP) {
        ft8rx_event e; e.msg_lo = lo; e.msg_hi = hi; e.cand = (uint16_t)cand; e.ipass = (uint8_t)ipass;
        e.slot = (uint8_t)slot; e.seq = (uint16_t)seq; e.valid = (uint16_t)valid;
        ev[(size_t)frame * FT8RX_EVENT_CAP + idx] = e;
    }
}

// ------------------------------------------------------------------------------------ event-log compaction (result copy of large batches)
// The log is [B][FT8RX_EVENT_CAP] x 24 B = 12 KB per frame with typically a tenth of it in use.  Before a large batch's results go
// to the host the used entries are packed back to back, frame after frame: k_ev_scan (one block) writes the exclusive prefix of
// min(count, cap) over the frames, k_ev_compact (a block per frame) moves the entries as dwords.
__global__ __launch_bounds__(1024) void k_ev_scan(const int32_t* __restrict__ evcount, int B, int32_t* __restrict__ offs) {
    __shared__ int s_wave[16];
    __shared__ int s_carry;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    if (tid == 0) s_carry = 0;
    __syncthreads();
    for (int base = 0; base < B; base += 1024) {
        const int f = base + tid;
        int c = f < B ? evcount[f] : 0;
        c = c > FT8RX_EVENT_CAP ? FT8RX_EVENT_CAP : (c < 0 ? 0 : c);
        int incl = c;                                                  // inclusive scan inside the wavefront
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) { const int t = __shfl_up(incl, d); if (lane >= d) incl += t; }
        if (lane == 63) s_wave[wv] = incl;
        __syncthreads();
        int before = s_carry;
        for (int i = 0; i < wv; i++) before += s_wave[i];
        if (f < B) offs[f] = before + incl - c;
        __syncthreads();
        if (tid == 1023) s_carry = before + incl;
        __syncthreads();
    }
    if (tid == 0) offs[B] = s_carry;
}
__global__ __launch_bounds__(64) void k_ev_compact(const ft8rx_event* __restrict__ ev, const int32_t* __restrict__ evcount,
                                                   const int32_t* __restrict__ offs, ft8rx_event* __restrict__ out) {
    const int f = blockIdx.x;
    int c = evcount[f]; c = c > FT8RX_EVENT_CAP ? FT8RX_EVENT_CAP : c;
    const uint32_t* src = reinterpret_cast<const uint32_t*>(ev + (size_t)f * FT8RX_EVENT_CAP);
    uint32_t* dst = reinterpret_cast<uint32_t*>(out + offs[f]);
    for (int i = threadIdx.x; i < c * 6; i += 64) dst[i] = src[i];
}

// ------------------------------------------------------------------------------------ packed results (multi-GPU gather, include/ft8rx.h)
// What the host message layer reads of a frame: the records of the candidates that decoded or made at least one unpack() call (an
// entry in the event log), and the used part of the log.  Three kernels at the end of a batch write exactly that into the caller's
// buffer: header | frame table | records | events.  k_pack_count (a block per frame) marks the candidates to keep, k_pack_scan (one
// block) turns the per-frame counts into offsets and the header, k_pack_write (a block per frame) moves the entries.
// A frame in which any candidate has a NaN llr_sd keeps ALL its candidates: the replay orders candidates by a stable sort on llr_sd
// (receiver.py:389), and with unordered keys the order of a subset need not be the order inside the full list.
static_assert(MAXC % 256 == 0 && sizeof(ft8rx_record) == 48 && sizeof(ft8rx_event) == 24 && sizeof(ft8rx_packed_frame) == 16 &&
              sizeof(ft8rx_packed_header) == 32, "packed result layout");
#define PK_NW (MAXC / 64)                       /* 64-candidate mask words per frame: 4 (libft8rx.so), 32 (wide build) */
__global__ __launch_bounds__(256) void k_pack_count(const ft8rx_record* __restrict__ rec, const int32_t* __restrict__ ncand,
                                                    const ft8rx_event* __restrict__ ev, const int32_t* __restrict__ evcount,
                                                    uint64_t* __restrict__ need /*[B][PK_NW]*/, int32_t* __restrict__ nrec /*[B]*/) {
    __shared__ uint32_t s_has[MAXC / 32];
    __shared__ int s_cnt[PK_NW], s_nan;
    const int f = blockIdx.x, tid = threadIdx.x, lane = tid & 63;
    for (int i = tid; i < MAXC / 32; i += 256) s_has[i] = 0;
    if (tid == 0) s_nan = 0;
    __syncthreads();
    int c = evcount[f]; c = c > FT8RX_EVENT_CAP ? FT8RX_EVENT_CAP : (c < 0 ? 0 : c);
    for (int i = tid; i < c; i += 256) {
        const unsigned cand = ev[(size_t)f * FT8RX_EVENT_CAP + i].cand;
        if (cand < (unsigned)MAXC) atomicOr(&s_has[cand >> 5], 1u << (cand & 31));
    }
    int n = ncand[f]; n = n > MAXC ? MAXC : (n < 0 ? 0 : n);
    for (int i = tid; i < n; i += 256) {
        const ft8rx_record& r = rec[(size_t)f * MAXC + i];
        if (r.grid_sd != r.grid_sd || r.fine_sd != r.fine_sd) s_nan = 1;
    }
    __syncthreads();
#pragma unroll 1
    for (int q = 0; q < MAXC / 256; q++) {
        const int i = tid + 256 * q;
        const bool decoded = i < n && rec[(size_t)f * MAXC + (i < n ? i : 0)].status == FT8RX_ST_DECODED;
        const bool want = i < n && (decoded || ((s_has[i >> 5] >> (i & 31)) & 1u) || s_nan);
        const uint64_t m = __ballot(want);
        if (lane == 0) { need[(size_t)f * PK_NW + (i >> 6)] = m; s_cnt[i >> 6] = __popcll(m); }
    }
    __syncthreads();
    if (tid == 0) { int t = 0; for (int w = 0; w < PK_NW; w++) t += s_cnt[w]; nrec[f] = t; }
}
// offsets of every frame's records / events in the packed runs, the frame table and the header (also mirrored into page-locked
// host memory, `hdr_host`, so that the host knows the size without a copy)
__global__ __launch_bounds__(1024) void k_pack_scan(const int32_t* __restrict__ nrec, const int32_t* __restrict__ ncand,
                                                    const int32_t* __restrict__ evcount, int B, int max_cands, unsigned long long cap_bytes,
                                                    unsigned char* __restrict__ buf, ft8rx_packed_header* __restrict__ hdr_host) {
    __shared__ int s_wr[16], s_we[16];
    __shared__ int s_cr, s_ce;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    ft8rx_packed_frame* table = reinterpret_cast<ft8rx_packed_frame*>(buf + sizeof(ft8rx_packed_header));
    const bool table_fits = sizeof(ft8rx_packed_header) + (size_t)B * sizeof(ft8rx_packed_frame) <= cap_bytes;
    if (tid == 0) { s_cr = 0; s_ce = 0; }
    __syncthreads();
    for (int base = 0; base < B; base += 1024) {
        const int f = base + tid;
        const int r = f < B ? nrec[f] : 0;
        const int eraw = f < B ? evcount[f] : 0;
        const int e = eraw > FT8RX_EVENT_CAP ? FT8RX_EVENT_CAP : (eraw < 0 ? 0 : eraw);
        int ir = r, ie = e;                                            // inclusive scans inside the wavefront
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) { const int tr = __shfl_up(ir, d), te = __shfl_up(ie, d); if (lane >= d) { ir += tr; ie += te; } }
        if (lane == 63) { s_wr[wv] = ir; s_we[wv] = ie; }
        __syncthreads();
        int br = s_cr, be = s_ce;
        for (int i = 0; i < wv; i++) { br += s_wr[i]; be += s_we[i]; }
        if (f < B && table_fits) {
            ft8rx_packed_frame t;
            t.rec_off = br + ir - r; t.ev_off = be + ie - e;
            int n = ncand[f]; n = n > MAXC ? MAXC : (n < 0 ? 0 : n);
            t.n_cand = (uint16_t)n; t.n_rec = (uint16_t)r; t.n_ev = eraw < 0 ? 0 : eraw;
            table[f] = t;
        }
        __syncthreads();
        if (tid == 1023) { s_cr = br + ir; s_ce = be + ie; }
        __syncthreads();
    }
    if (tid == 0) {
        ft8rx_packed_header hd;
        hd.magic = FT8RX_PACKED_MAGIC; hd.n_frames = B; hd.n_records = s_cr; hd.n_events = s_ce; hd.max_cands = max_cands;
        hd.bytes = sizeof(ft8rx_packed_header) + (unsigned long long)B * sizeof(ft8rx_packed_frame) +
                   (unsigned long long)s_cr * sizeof(ft8rx_record) + (unsigned long long)s_ce * sizeof(ft8rx_event);
        hd.overflow = hd.bytes > cap_bytes;
        if (cap_bytes >= sizeof(hd)) *reinterpret_cast<ft8rx_packed_header*>(buf) = hd;
        *hdr_host = hd;
    }
}
__global__ __launch_bounds__(256) void k_pack_write(const ft8rx_record* __restrict__ rec, const ft8rx_event* __restrict__ ev,
                                                    const uint64_t* __restrict__ need, int B, unsigned char* __restrict__ buf) {
    const ft8rx_packed_header hd = *reinterpret_cast<const ft8rx_packed_header*>(buf);
    if (hd.overflow) return;
    __shared__ int s_pre[PK_NW];                                       // kept records in the mask words before word w
    const int f = blockIdx.x, tid = threadIdx.x, lane = tid & 63;
    const ft8rx_packed_frame t = reinterpret_cast<const ft8rx_packed_frame*>(buf + sizeof(ft8rx_packed_header))[f];
    unsigned char* recs = buf + sizeof(ft8rx_packed_header) + (size_t)B * sizeof(ft8rx_packed_frame);
    unsigned char* evs = recs + (size_t)hd.n_records * sizeof(ft8rx_record);
    if (tid == 0) { int a = 0; for (int w = 0; w < PK_NW; w++) { s_pre[w] = a; a += __popcll(need[(size_t)f * PK_NW + w]); } }
    __syncthreads();
#pragma unroll 1
    for (int q = 0; q < MAXC / 256; q++) {
        const int i = tid + 256 * q;
        const uint64_t mine = need[(size_t)f * PK_NW + (i >> 6)];
        if ((mine >> lane) & 1ull) {
            const int p = s_pre[i >> 6] + __popcll(mine & ((1ull << lane) - 1));
            const uint4* src = reinterpret_cast<const uint4*>(rec + (size_t)f * MAXC + i);
            uint4* dst = reinterpret_cast<uint4*>(recs + ((size_t)t.rec_off + p) * sizeof(ft8rx_record));
            uint4 a = src[0], b = src[1], c = src[2];
            c.w = (uint32_t)i;                                         // pad2 = the candidate's index inside its frame
            dst[0] = a; dst[1] = b; dst[2] = c;
        }
    }
    const int ne = t.n_ev > FT8RX_EVENT_CAP ? FT8RX_EVENT_CAP : t.n_ev;
    const uint2* es = reinterpret_cast<const uint2*>(ev + (size_t)f * FT8RX_EVENT_CAP);
    uint2* ed = reinterpret_cast<uint2*>(evs + (size_t)t.ev_off * sizeof(ft8rx_event));
    for (int i = tid; i < ne * 3; i += 256) ed[i] = es[i];
}
#endif  // FT8RX_ILP_UNIT

#endif
