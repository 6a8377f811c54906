// ft8rx.hip -- MI355X (gfx950) FT8 receive hot path: HIP kernels + the C ABI of include/ft8rx.h.
//
// Pipeline for a batch of B independent 15-s frames (all stream-ordered on one HIP stream, no host
// round trips; frames are the batch dimension, candidates the second one):
//   k_spectrogram  (hop, frame)         Hann * 3840-pt real FFT (1920-pt complex Stockham in LDS) -> dB grid
//   k_sync         (16-f0 tile, frame)  Costas correlation over all time offsets from an LDS tile
//   k_topk         (frame)              threshold + stable top-K (bitonic sort in LDS)
//   k_grid_llr     (candidate)          payload gather -> max-log LLRs -> sigma normalisation
//   k_bp           (candidate, AP)      one wavefront: GOOD91 + flooding BP, ballot parity, LDS exchange
//   k_select0      (candidate)          first success in ladder order
//   k_cyc_a/b/c    (tile, frame)        192000-pt real FFT as 300x320 four-step + real split
//   k_fine         (candidate)          9x(slice/taper/3200-pt IFFT) + 32-pt DFT scoring, Costas gate, LLRs
//   k_bp           (candidate, AP)      GOOD91 + BP(90,20) with saved outputs
//   k_select1, k_osd (candidate, slot) one wavefront: rank sort, register-resident GF(2) Gauss-Jordan
//                                       with ballot pivoting, lane-per-trial CRC-14 + validity, k_select2
// Reference line citations are to PyFT8/receiver.py and PyFT8/decoders.py (see include/ft8rx.h).
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <string>
#include <vector>
#include "../../include/ft8rx.h"
#include "ft8_dev.h"

#define MAXC FT8RX_MAX_CANDS
#define NF0MAX 1024

// ------------------------------------------------------------------------------------ device tables
struct Tables {
    const float* win;        // [3840] Hann (np.hanning) as f32
    const cpx* W1920;        // twiddles
    const cpx* WR3840;       // [976] real-split twiddles e^{-2 pi i k/3840}
    const cpx* W3200;
    const cpx* W96000;
    const cpx* W300;
    const cpx* W320;
    const cpx* WR192k;       // [49152]
    const cpx* W32;
    const double* taper;     // [100]
};

__device__ __constant__ int d_COSTAS[7] = {3, 1, 4, 0, 6, 5, 2};
__device__ __constant__ uint8_t d_PAYSYM[58] = {7,8,9,10,11,12,13,14,15,16,17,18,19,20,21,22,23,24,25,26,27,28,29,30,31,32,33,34,35,
    43,44,45,46,47,48,49,50,51,52,53,54,55,56,57,58,59,60,61,62,63,64,65,66,67,68,69,70,71};
// AP masks (reference receiver.py:21-27), copied verbatim as data
__device__ __constant__ int8_t d_AP_CQ[29]   = {0,0,0,0,0, 0,0,0,0,0, 0,0,0,0,0, 0,0,0,0,0, 0,0,0,0,0, 0,1,0,0};
__device__ __constant__ int8_t d_AP_END[3][19] = {{0,1, 1,1,1,1,1, 0,0,1,1,1, 0,1,0,1,0, 0,1},
                                                  {0,1, 1,1,1,1,1, 0,1,0,0,1, 0,1,0,0,0, 0,1},
                                                  {0,1, 1,1,1,1,1, 0,1,0,0,1, 0,0,1,0,0, 0,1}};
// LDPC tables in device memory (copies of ft8_tables.h)
__device__ uint8_t  d_CHK_N[83];
__device__ int16_t  d_CHK_V[83][7];
__device__ uint16_t d_CHK_E0[83];
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

#define W6 ((double)(-0.16666667163372040f))     /* np.float32(-1/6), receiver.py:323,198 */

// grid row accessor with the reference's modulo-750 wrap (receiver.py:240,347,360)
FT8_DEV float grid_at(const float* __restrict__ g, int row, int col) {
    row %= 750; if (row < 0) row += 750;
    if (row >= 1 && row <= 375) return g[row * FT8RX_GRID_COLS + col];
    return 1.0f;
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

// ------------------------------------------------------------------------------------ K1 spectrogram
// one hop: window samples a[base .. base+3840) (zeros before the frame start) -> 976 dB values.
// 128 threads; 1920-point complex FFT (plan [8,4,4,5,3]) in place in one LDS image as three register-fused stages:
//   [8]    240 butterflies straight from global memory (int16 -> f32, Hann window fused in),
//   [4,4]  120 groups of 16 (one per thread), twiddles from the LDS table w240[t] = W1920[8 t],
//   [5,3]  128 groups of 15 (one per thread), compile-time twiddles,
// then the real-FFT split and 20 log10|.|.
#define SPEC_NT 128
FT8_DEV void spectrogram_hop(const int16_t* __restrict__ a, int base, float* __restrict__ out, const Tables& T,
                             cpx* z, cpx* w240, int tid) {
    const cpx* __restrict__ W = T.W1920;
    for (int i = tid; i < 240; i += SPEC_NT) w240[i] = W[8 * i];
    {   // pass [8]: n = 1920, s = 1, m = 240: butterfly p reads samples m = p + 240 j
        cpx v[2][8];
#pragma unroll
        for (int i = 0; i < 2; i++) {
            const int p = tid + SPEC_NT * i;
#pragma unroll
            for (int j = 0; j < 8; j++) {
                const int m = p + 240 * j, i0 = base + 2 * m;
                float x0 = 0.0f, x1 = 0.0f;
                if (p < 240 && i0 >= 0) {
                    const short2 sm = *reinterpret_cast<const short2*>(a + i0);
                    const float2 w = *reinterpret_cast<const float2*>(T.win + 2 * m);
                    x0 = (float)sm.x * w.x; x1 = (float)sm.y * w.y;
                }
                v[i][j] = make_float2(x0, x1);
            }
        }
#pragma unroll
        for (int i = 0; i < 2; i++) {
            const int p = tid + SPEC_NT * i;
            if (p < 240) {
                dft<8>(v[i]);
                z[8 * p] = v[i][0];
#pragma unroll
                for (int j = 1; j < 8; j++) { cpx t = v[i][j]; if (p != 0) t = cmul(t, W[j * p]); z[8 * p + j] = t; }
            }
        }
    }
    __syncthreads();
    {   // passes [4,4]: n = 240, s = 8; group g = (pp = g / 8, q = g % 8): in q + 8(pp + 15 j' + 60 j), out q + 8 j + 32(4 pp + j')
        typedef Fused2<1920, 240, 8, 4, 4> F;
        cpx v[4][4];
        const bool on = tid < F::groups;
        if (on) F::load_affine<120, 480>(z, tid, v);
        __syncthreads();
        if (on) {
            const int pp = tid >> 3;
#pragma unroll
            for (int jp = 0; jp < 4; jp++) {
                dft<4>(v[jp]);
                const int pq = pp + 15 * jp;
                if (pq != 0) {
#pragma unroll
                    for (int j = 1; j < 4; j++) v[jp][j] = cmul(v[jp][j], w240[j * pq]);          // W1920[j p 8]
                }
            }
#pragma unroll
            for (int j = 0; j < 4; j++) {
                cpx u[4];
#pragma unroll
                for (int jp = 0; jp < 4; jp++) u[jp] = v[jp][j];
                dft<4>(u);
                if (pp != 0) {
#pragma unroll
                    for (int jp = 1; jp < 4; jp++) u[jp] = cmul(u[jp], w240[4 * jp * pp]);        // W1920[j' pp 32]
                }
#pragma unroll
                for (int jp = 0; jp < 4; jp++) v[jp][j] = u[jp];
            }
            F::store_affine<32, 8>(z, (tid & 7) + 128 * (tid >> 3), v);
        }
        __syncthreads();
    }
    {   // passes [5,3]: n = 15, s = 128; group q: in q + 128 (j' + 3 j), out q + 128 j + 640 j'
        typedef Fused2<1920, 15, 128, 5, 3> F;
        cpx v[3][5];
        F::load_affine<128, 384>(z, tid, v);
        __syncthreads();
        F::compute_pp(0, v, W);
        F::store_affine<640, 128>(z, tid, v);
        __syncthreads();
    }
    for (int k = tid; k < FT8RX_GRID_COLS; k += SPEC_NT) {
        cpx p = z[k], q = z[(1920 - k) % 1920];
        float er = 0.5f * (p.x + q.x), ei = 0.5f * (p.y - q.y);
        float orr = 0.5f * (p.y + q.y), oi = 0.5f * (q.x - p.x);
        cpx w = T.WR3840[k];
        float xr = er + (w.x * orr - w.y * oi);
        float xi = ei + (w.x * oi + w.y * orr);
        float mag = sqrtf(xr * xr + xi * xi);
        out[k] = 20.0f * ft8_log10f(mag + 1e-12f);
    }
}

__global__ __launch_bounds__(SPEC_NT) void k_spectrogram(const int16_t* __restrict__ audio, float* __restrict__ grid, Tables T) {
    __shared__ cpx z[1920];
    __shared__ cpx w240[240];
    // XCD-aware hop mapping: workgroup id -> XCD is id % 8 and gridDim.x = 376 = 8 * 47, so the 47 workgroups of a
    // frame that land on one XCD take 47 consecutive hops: each XCD's L2 then sees one eighth of the frame's audio
    // (8x overlapping windows) instead of all of it.
    const int hop = (blockIdx.x & 7) * 47 + (blockIdx.x >> 3) + 1, f = blockIdx.y, tid = threadIdx.x;
    if (hop > 375) return;
    spectrogram_hop(audio + (size_t)f * FT8RX_NSAMP, 480 * hop - 3840,
                    grid + ((size_t)f * FT8RX_GRID_ROWS + hop) * FT8RX_GRID_COLS, T, z, w240, tid);
}

// streaming mode: one hop of the live receiver (AudioIn.get_hop_spectrum, receiver.py:288-293)
__global__ __launch_bounds__(SPEC_NT) void k_hop_spectrum(const int16_t* __restrict__ win3840, float* __restrict__ row, Tables T) {
    __shared__ cpx z[1920];
    __shared__ cpx w240[240];
    spectrogram_hop(win3840, 0, row, T, z, w240, threadIdx.x);
}

__global__ void k_fill_row0(float* grid, int B) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < B * FT8RX_GRID_COLS) grid[(size_t)(i / FT8RX_GRID_COLS) * FT8RX_GRID_ROWS * FT8RX_GRID_COLS + (i % FT8RX_GRID_COLS)] = 1.0f;
}

// ------------------------------------------------------------------------------------ K2 sync search
// block = 16 consecutive f0 of one frame; LDS tile = every grid row any h0 can touch x 29 columns.
__global__ __launch_bounds__(256) void k_sync(const float* __restrict__ grid, float* __restrict__ best_score,
                                              int32_t* __restrict__ best_h0, ft8rx_config cfg) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int nh0 = cfg.h0_hi - cfg.h0_lo;
    const int nrows = nh0 + 24;
    double* T = reinterpret_cast<double*>(smem);                   // [nrows][16] 14-bin window sums (fp64)
    float* tile = reinterpret_cast<float*>(T + nrows * 16);       // [nrows][29]
    float* redS = tile + nrows * 29;                              // [256]
    int* redH = reinterpret_cast<int*>(redS + 256);               // [256]
    const int f = blockIdx.y, tid = threadIdx.x;
    const int f0base = cfg.f0_lo + 16 * blockIdx.x;
    const int rlo = cfg.h0_lo + 148;
    const float* g = grid + (size_t)f * FT8RX_GRID_ROWS * FT8RX_GRID_COLS;
    for (int i = tid; i < nrows * 29; i += 256) {
        int r = i / 29, c = i - r * 29;
        int col = f0base + c;
        tile[i] = (col < FT8RX_GRID_COLS) ? grid_at(g, rlo + r, col) : 0.0f;
    }
    __syncthreads();
    // T[r][f] = sum_{b<14} tile[r][f+b], accumulated in the contract's order (b ascending, fp64); every time offset that
    // touches row r reuses it, so the 98-tap correlation becomes 7 window sums + 14 tone-bin reads.
    for (int i = tid; i < nrows * 16; i += 256) {
        const float* row = tile + (i >> 4) * 29 + (i & 15);
        double t = 0.0;
#pragma unroll
        for (int b = 0; b < 14; b++) t += (double)row[b];
        T[i] = t;
    }
    __syncthreads();
    const int f0l = tid & 15;
    float best = 0.0f; int bh = 0;
    for (int hi = tid >> 4; hi < nh0; hi += 16) {
        double s1 = 0.0, tsum = 0.0;
#pragma unroll
        for (int s = 0; s < 7; s++) {
            const int r = hi + 4 * s;
            const float* row = tile + r * 29 + f0l;
            tsum += T[r * 16 + f0l];
            const int c = d_COSTAS[s];
            s1 += (double)row[2 * c] + (double)row[2 * c + 1];
        }
        float score = (float)(s1 + W6 * (tsum - s1));
        if (score > best) { best = score; bh = cfg.h0_lo + hi; }      // ascending h0 => first strict maximum
    }
    redS[tid] = best; redH[tid] = bh;
    __syncthreads();
    if (tid < 16) {
        float bs = 0.0f; int h = 0;
        for (int gI = 0; gI < 16; gI++) {
            float s = redS[tid + 16 * gI]; int hh = redH[tid + 16 * gI];
            if (s > bs || (s == bs && s > 0.0f && hh < h)) { bs = s; h = hh; }
        }
        int f0 = f0base + tid;
        if (f0 < cfg.f0_hi) {
            best_score[(size_t)f * NF0MAX + (f0 - cfg.f0_lo)] = bs;
            best_h0[(size_t)f * NF0MAX + (f0 - cfg.f0_lo)] = h;
        }
    }
}

// ------------------------------------------------------------------------------------ K3 top-K
// threshold, stable sort by score descending (ties: f0 ascending = original order), keep max_cands
__global__ __launch_bounds__(1024) void k_topk(const float* __restrict__ best_score, const int32_t* __restrict__ best_h0,
                                               ft8rx_record* __restrict__ rec, int32_t* __restrict__ ncand, ft8rx_config cfg) {
    __shared__ uint64_t key[1024];
    const int f = blockIdx.x, tid = threadIdx.x;
    const int nf0 = cfg.f0_hi - cfg.f0_lo;
    uint64_t k = ~0ull;
    if (tid < nf0) {
        float s = best_score[(size_t)f * NF0MAX + tid];
        if (s > cfg.sync_score_min) k = ((uint64_t)(~__float_as_uint(s)) << 32) | (uint32_t)tid;   // s > 0: bit pattern is monotonic
    }
    key[tid] = k;
    __syncthreads();
    for (int size = 2; size <= 1024; size <<= 1) {
        for (int stride = size >> 1; stride > 0; stride >>= 1) {
            int partner = tid ^ stride;
            if (partner > tid) {
                uint64_t a = key[tid], b = key[partner];
                bool up = ((tid & size) == 0);
                if ((a > b) == up) { key[tid] = b; key[partner] = a; }
            }
            __syncthreads();
        }
    }
    uint64_t kk = key[tid];
    int cnt = __syncthreads_count(kk != ~0ull);
    if (tid == 0) ncand[f] = cnt < cfg.max_cands ? cnt : cfg.max_cands;
    if (tid < cfg.max_cands) {
        ft8rx_record r; memset(&r, 0, sizeof(r));
        if (kk != ~0ull) {
            int i = (int)(kk & 0xffffffffu);
            r.f0_idx = (int16_t)(cfg.f0_lo + i);
            r.h0_idx = (int16_t)best_h0[(size_t)f * NF0MAX + i];
            r.score = __uint_as_float(~(uint32_t)(kk >> 32));
            r.status = FT8RX_ST_ACTIVE; r.ipass = 0xff;
        } else r.status = FT8RX_ST_EXHAUSTED;
        rec[(size_t)f * MAXC + tid] = r;
    }
}

// ------------------------------------------------------------------------------------ LLR extraction (receiver.py:208-222)
// p[464] dB values in LDS -> normalised llr[174] in LDS `llr`.  Every thread of the block must call this
// (it contains block barriers); only the threads with active==true (exactly one wavefront, lane = its
// lane id) do the work.  sd/snr are returned to the active lanes.
FT8_DEV void llr_from_p(const float* p, float* llr, float* sq, int lane, bool active, float* sd_out, int* snr_out) {
    float sd = 0.0f; int snr = 0;
    if (active) {
        float pmax = -__builtin_inff(), pmin = __builtin_inff();
        for (int i = lane; i < 464; i += 64) { float v = p[i]; if (v > pmax) pmax = v; if (v < pmin) pmin = v; }
        for (int o = 32; o > 0; o >>= 1) {
            float a = __shfl_xor(pmax, o), b = __shfl_xor(pmin, o);
            if (a > pmax) pmax = a;
            if (b < pmin) pmin = b;
        }
        float d = (pmax - pmin) - 58.0f;
        snr = (int)d; if (snr < -24) snr = -24; if (snr > 24) snr = 24;
        if (lane < 58) {
            const float* q = p + 8 * lane;
            float q0 = q[0], q1 = q[1], q2 = q[2], q3 = q[3], q4 = q[4], q5 = q[5], q6 = q[6], q7 = q[7];
#define MAX4(a, b, c, d) ({ float _m = (a); if ((b) > _m) _m = (b); if ((c) > _m) _m = (c); if ((d) > _m) _m = (d); _m; })
            float la = MAX4(q4, q5, q6, q7) - MAX4(q0, q1, q2, q3);
            float lb = MAX4(q2, q3, q4, q7) - MAX4(q0, q1, q5, q6);
            float lc = MAX4(q1, q2, q6, q7) - MAX4(q0, q3, q4, q5);
#undef MAX4
            llr[3 * lane] = la; llr[3 * lane + 1] = lb; llr[3 * lane + 2] = lc;
            sq[3 * lane] = la * la; sq[3 * lane + 1] = lb * lb; sq[3 * lane + 2] = lc * lc;
        }
    }
    __syncthreads();
    if (active) {
        // numpy pairwise float32 sums of llr (lanes 0..15) and llr^2 (lanes 16..31): n=174 -> blocks [0,80) and [80,174)
        const float* arr = (lane & 16) ? sq : llr;
        const int j = lane & 7, half = (lane >> 3) & 1;
        const int base = half ? 80 : 0, nblk = half ? 88 : 80;
        float r = arr[base + j];
        for (int i = 8; i < nblk; i += 8) r += arr[base + i + j];
        r = r + __shfl_xor(r, 1);
        r = r + __shfl_xor(r, 2);
        r = r + __shfl_xor(r, 4);
        if (half) for (int i = 88; i < 94; i++) r += arr[80 + i];
        float tot_l = __shfl(r, 0) + __shfl(r, 8);
        float tot_s = __shfl(r, 16) + __shfl(r, 24);
        float mean = tot_l / 174.0f;
        float var = tot_s / 174.0f - mean * mean;
        sd = sqrtf(var);
    }
    __syncthreads();
    if (active) for (int i = lane; i < 174; i += 64) llr[i] = (2.83f * llr[i]) / sd;
    __syncthreads();
    *sd_out = sd; *snr_out = snr;
}

// block of 64 = one candidate (or one test triple when `trip` is given)
__global__ __launch_bounds__(64) void k_grid_llr(const float* __restrict__ grid, ft8rx_record* __restrict__ rec,
                                                 const int32_t* __restrict__ ncand, float* __restrict__ llr0,
                                                 ft8rx_config cfg, const int32_t* __restrict__ trip, float* __restrict__ t_sd,
                                                 int32_t* __restrict__ t_snr) {
    __shared__ float p[464];
    __shared__ float llr[174];
    __shared__ float sq[174];
    const int lane = threadIdx.x;
    int frame, ci, f0, h0;
    if (trip) { frame = trip[3 * blockIdx.x]; f0 = trip[3 * blockIdx.x + 1]; h0 = trip[3 * blockIdx.x + 2]; ci = 0; }
    else {
        frame = blockIdx.x / MAXC; ci = blockIdx.x % MAXC;
        if (ci >= ncand[frame]) return;
        const ft8rx_record& r = rec[(size_t)frame * MAXC + ci];
        f0 = r.f0_idx; h0 = r.h0_idx;
    }
    const float* g = grid + (size_t)frame * FT8RX_GRID_ROWS * FT8RX_GRID_COLS;
    for (int i = lane; i < 464; i += 64) {
        int s = i >> 3, t = i & 7;
        p[i] = grid_at(g, h0 + 4 + 4 * (int)d_PAYSYM[s], f0 + 1 + 2 * t);        // receiver.py:358-362
    }
    __syncthreads();
    float sd; int snr;
    llr_from_p(p, llr, sq, lane, true, &sd, &snr);
    float* out = llr0 + (size_t)blockIdx.x * 174;
    for (int i = lane; i < 174; i += 64) out[i] = llr[i];
    if (lane == 0) {
        if (trip) { t_sd[blockIdx.x] = sd; t_snr[blockIdx.x] = snr; }
        else {
            ft8rx_record& r = rec[(size_t)frame * MAXC + ci];
            r.grid_sd = sd; r.snr_grid = (int8_t)snr;
            if (sd <= cfg.llr_sd_min) r.status = FT8RX_ST_STOP_GRID_SD;
        }
    }
}

// ------------------------------------------------------------------------------------ AP masks (receiver.py:109-117)
FT8_DEV float ap_value(int ap, int i, float v) {
    if (ap == 1) {
        if (i < 29) return d_AP_CQ[i] ? 5.0f : -5.0f;
        if (i == 74 || i == 75 || i == 57 || i == 58) return -5.0f;
        if (i == 76) return 5.0f;
    } else if (ap >= 2) {
        if (i >= 58 && i < 77) return d_AP_END[ap - 2][i - 58] ? 5.0f : -5.0f;
    }
    return v;
}

// ------------------------------------------------------------------------------------ LDPC belief propagation
// One wavefront per (candidate, AP) -- or per test vector.  Edge-parallel tanh / message update
// (lane l owns edges l, l+64, ...), check-parallel products, variable-parallel accumulation in the
// reference's np.add.at order; everything exchanged through LDS; parity via __ballot.
// mode 0: pipeline ipass 0 (GOOD91 then BP(nc0_a, iters_a)), mode 1: pipeline fine stage
// (GOOD91 for ap<2, BP(nc0_b, iters_b), save output llr), mode 2: raw vectors (tests).
__global__ __launch_bounds__(64) void k_bp(int mode, const float* __restrict__ llr_in, ft8rx_record* __restrict__ rec,
                                           const int32_t* __restrict__ ncand, Att* __restrict__ attG, Att* __restrict__ attB,
                                           float* __restrict__ saved, ft8rx_event* ev, int32_t* evcount, ft8rx_config cfg,
                                           int max_nc0, int max_iters) {
    __shared__ float llr[176];
    __shared__ float tl[528];
    __shared__ float dl[528];
    __shared__ float P[84];
    const int lane = threadIdx.x;
    int frame = 0, ci = 0, ap = 0; size_t vec;
    if (mode == 2) vec = blockIdx.x;
    else {
        ap = blockIdx.x % 5; int c = blockIdx.x / 5; frame = c / MAXC; ci = c % MAXC;
        if (ci >= ncand[frame]) return;
        if (rec[(size_t)frame * MAXC + ci].status != FT8RX_ST_ACTIVE) return;
        vec = (size_t)c;
    }
    for (int i = lane; i < 174; i += 64) llr[i] = ap_value(ap, i, llr_in[vec * 174 + i]);
    __syncthreads();
    Att res; memset(&res, 0, sizeof(res)); res.n_its = -1;
    Att resG; memset(&resG, 0, sizeof(resG)); resG.n_its = -1;
    const int ipG = (mode == 0) ? 0 : 2;
    // ---- GOOD91: CRC on the hard decisions of llr[:91] (receiver.py:119-122)
    bool doneG = false;
    if (mode == 0 || (mode == 1 && ap < 2)) {
        uint64_t b0 = __ballot(llr[lane] > 0.0f);
        uint64_t b1 = __ballot(lane < 27 && llr[64 + (lane < 27 ? lane : 0)] > 0.0f);
        uint64_t lo, hi;
        int r = ft8_crc_check(b0, b1, &lo, &hi);
        if (r) { if (lane == 0) log_event(ev, evcount, frame, ci, ipG, ap, 0, lo, hi, r == 2); }
        if (r == 2) { resG.ok = 1; resG.lo = lo; resG.hi = hi; resG.n_its = 0; resG.method = FT8RX_M_GOOD91; doneG = true; }
    }
    // membership masks of this lane's two checks (c0 = lane, c1 = 64 + lane) over the 174 variables
    const int c0 = lane, c1 = lane + 64;
    const uint64_t cm00 = d_CHK_MASK[c0][0], cm01 = d_CHK_MASK[c0][1], cm02 = d_CHK_MASK[c0][2];
    const uint64_t cm10 = d_CHK_MASK[c1][0], cm11 = d_CHK_MASK[c1][1], cm12 = d_CHK_MASK[c1][2];
    // the edge tables are only needed once BP really iterates: most ipass-0 attempts stop at the initial
    // unsatisfied-check test (decoders.py:159), so they are loaded lazily below
    int ev_[9], ec_[9];
    int n0 = 0, e00 = 0, n1 = 0, e01 = 0;
    bool tables = false;
    float mc[9];
#pragma unroll
    for (int i = 0; i < 9; i++) mc[i] = 0.0f;
    bool run_bp = !(mode == 0 && doneG);       // ipass 0: the BP of this AP is only reached if GOOD91 failed
    res.has_out = 1;
    if (run_bp) for (int it = 0; it < max_iters; it++) {
        // parity of every check from the hard decisions
        const uint64_t h0 = __ballot(llr[lane] > 0.0f), h1 = __ballot(llr[64 + lane] > 0.0f),
                       h2 = __ballot(lane < 46 && llr[128 + (lane < 46 ? lane : 0)] > 0.0f);
        const int par0 = (__popcll(h0 & cm00) + __popcll(h1 & cm01) + __popcll(h2 & cm02)) & 1;
        const int par1 = (__popcll(h0 & cm10) + __popcll(h1 & cm11) + __popcll(h2 & cm12)) & 1;
        int ncheck = __popcll(__ballot(par0)) + __popcll(__ballot(par1));
        if (it == 0) { res.nc0 = (uint8_t)ncheck; if (ncheck > max_nc0) { res.has_out = 0; break; } }
        if (ncheck == 0) {
            uint64_t b0 = h0;
            uint64_t b1 = h1 & ((1ull << 27) - 1);
            uint64_t lo, hi;
            int r = ft8_crc_check(b0, b1, &lo, &hi);
            if (r) {
                int ipass = (mode == 0) ? 0 : ((ap < 2 && res.nc0 <= cfg.bp_nc0_a && it < cfg.bp_iters_a) ? 3 : 4);
                if (lane == 0) log_event(ev, evcount, frame, ci, ipass, ap, it + 1, lo, hi, r == 2);
            }
            if (r == 2) { res.ok = 1; res.lo = lo; res.hi = hi; res.n_its = (int16_t)it; res.has_out = 0; }
            break;      // success, or frozen state: the reference changes nothing from here on (decoders.py:161-164)
        }
        if (!tables) {                 // wave-uniform: first real iteration
            tables = true;
#pragma unroll
            for (int i = 0; i < 9; i++) { int e = lane + 64 * i; ev_[i] = (e < 522) ? d_EDGE_V[e] : 0; ec_[i] = (e < 522) ? d_EDGE_C[e] : 0; }
            n0 = d_CHK_N[c0]; e00 = d_CHK_E0[c0];
            n1 = (c1 < 83) ? d_CHK_N[c1] : 0; e01 = (c1 < 83) ? d_CHK_E0[c1] : 0;
        }
        float tt[9];
#pragma unroll
        for (int i = 0; i < 9; i++) {
            int e = lane + 64 * i;
            if (e < 522) { float v2c = llr[ev_[i]] - mc[i]; tt[i] = ft8_tanhf(-v2c); tl[e] = tt[i]; }
        }
        __syncthreads();
        {
            float Pp = tl[e00];
#pragma unroll
            for (int j = 1; j < 6; j++) Pp = Pp * tl[e00 + j];
            if (n0 == 7) Pp = Pp * tl[e00 + 6];
            P[c0] = Pp;
            if (c1 < 83) {
                float Q = tl[e01];
#pragma unroll
                for (int j = 1; j < 6; j++) Q = Q * tl[e01 + j];
                if (n1 == 7) Q = Q * tl[e01 + 6];
                P[c1] = Q;
            }
        }
        __syncthreads();
#pragma unroll
        for (int i = 0; i < 9; i++) {
            int e = lane + 64 * i;
            if (e < 522) {
                const float Pc = P[ec_[i]], u = 1.18f * tt[i];
                float nm = (Pc * tt[i]) / ((Pc - u) * (u + Pc));     // = e/((e-1.18)(1.18+e)), e = P/t, in one division
                dl[e] = nm - mc[i];
                mc[i] = nm;
            }
        }
        __syncthreads();
        for (int v = lane; v < 174; v += 64) {
            float col = 0.0f;
            col += dl[d_VAR_E[v][0]]; col += dl[d_VAR_E[v][1]]; col += dl[d_VAR_E[v][2]];
            llr[v] += col;
        }
        __syncthreads();
    }
    else res.has_out = 0;
    if (res.ok) res.method = (mode == 0) ? FT8RX_M_LDPC_A : FT8RX_M_LDPC_B;
    if (mode == 2) {
        if (lane == 0) attB[vec] = res;
        if (res.has_out) for (int i = lane; i < 174; i += 64) saved[vec * 174 + i] = llr[i];
        return;
    }
    if (mode == 0) { if (lane == 0) attB[vec * 5 + ap] = doneG ? resG : res; return; }
    if (lane == 0) { attB[vec * 5 + ap] = res; if (ap < 2) attG[vec * 2 + ap] = resG; }
    if (res.has_out) for (int i = lane; i < 174; i += 64) saved[(vec * 5 + ap) * 174 + i] = llr[i];
}

// first success in ladder order after ipass 0 (receiver.py:72-78)
__global__ void k_select0(ft8rx_record* rec, const int32_t* ncand, const Att* att0, int B) {
    int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= B * MAXC) return;
    int frame = c / MAXC, ci = c % MAXC;
    if (ci >= ncand[frame]) return;
    ft8rx_record& r = rec[c];
    if (r.status != FT8RX_ST_ACTIVE) return;
    for (int ap = 0; ap < 5; ap++) {
        const Att& a = att0[(size_t)c * 5 + ap];
        if (a.ok) { r.status = FT8RX_ST_DECODED; r.ipass = 0; r.ap = (uint8_t)ap; r.method = a.method; r.n_its = a.n_its; r.msg_lo = a.lo; r.msg_hi = a.hi; return; }
    }
}

// first success among ipass 2 (GOOD91 ap0,1), 3 (BP_A ap0,1 derived from the BP_B run), 4 (BP_B ap0..4)
__global__ void k_select1(ft8rx_record* rec, const int32_t* ncand, const Att* attG, const Att* attB, int B, ft8rx_config cfg) {
    int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= B * MAXC) return;
    int frame = c / MAXC, ci = c % MAXC;
    if (ci >= ncand[frame]) return;
    ft8rx_record& r = rec[c];
    if (r.status != FT8RX_ST_ACTIVE) return;
    for (int ap = 0; ap < 2; ap++) {
        const Att& a = attG[(size_t)c * 2 + ap];
        if (a.ok) { r.status = FT8RX_ST_DECODED; r.ipass = 2; r.ap = (uint8_t)ap; r.method = FT8RX_M_GOOD91; r.n_its = 0; r.msg_lo = a.lo; r.msg_hi = a.hi; return; }
    }
    for (int ap = 0; ap < 2; ap++) {
        const Att& a = attB[(size_t)c * 5 + ap];
        if (a.ok && a.nc0 <= cfg.bp_nc0_a && a.n_its < cfg.bp_iters_a) {
            r.status = FT8RX_ST_DECODED; r.ipass = 3; r.ap = (uint8_t)ap; r.method = FT8RX_M_LDPC_A; r.n_its = a.n_its; r.msg_lo = a.lo; r.msg_hi = a.hi; return; }
    }
    for (int ap = 0; ap < 5; ap++) {
        const Att& a = attB[(size_t)c * 5 + ap];
        if (a.ok) { r.status = FT8RX_ST_DECODED; r.ipass = 4; r.ap = (uint8_t)ap; r.method = FT8RX_M_LDPC_B; r.n_its = a.n_its; r.msg_lo = a.lo; r.msg_hi = a.hi; return; }
    }
}

// ipass 5 (OSD on llr0+AP, slots 0..4) then ipass 6 (OSD on the saved BP outputs, slots 5..9)
__global__ void k_select2(ft8rx_record* rec, const int32_t* ncand, const Att* attO, int B) {
    int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= B * MAXC) return;
    int frame = c / MAXC, ci = c % MAXC;
    if (ci >= ncand[frame]) return;
    ft8rx_record& r = rec[c];
    if (r.status != FT8RX_ST_ACTIVE) return;
    for (int s = 0; s < 10; s++) {
        const Att& a = attO[(size_t)c * 10 + s];
        if (a.ok) {
            r.status = FT8RX_ST_DECODED; r.ipass = (s < 5) ? 5 : 6; r.ap = (uint8_t)(s % 5);
            r.method = (s < 5) ? FT8RX_M_OSD : FT8RX_M_LDPC_B_OSD; r.n_its = a.n_its; r.msg_lo = a.lo; r.msg_hi = a.hi; return; }
    }
    r.status = FT8RX_ST_EXHAUSTED;
}

// ------------------------------------------------------------------------------------ cycle spectrum: 192000-pt real FFT
// z[m] = x[2m] + i x[2m+1], 96000 = 300 x 320 four-step, then the real split for bins < 49152.
__global__ __launch_bounds__(256) void k_cyc_a(const int16_t* __restrict__ audio, cpx* __restrict__ A, Tables T) {
    __shared__ cpx bufA[8 * 300];
    __shared__ cpx bufB[8 * 300];
    // XCD-aware tile mapping (workgroup id % 8 = XCD, gridDim.x = 40 = 8 * 5): one XCD takes 5 adjacent column tiles,
    // i.e. 160 contiguous bytes of every audio row, so the 128-B lines are shared inside one L2 instead of four.
    const int tile = (blockIdx.x & 7) * 5 + (blockIdx.x >> 3);
    const int f = blockIdx.y, tid = threadIdx.x, n2b = 8 * tile;
    const int16_t* a = audio + (size_t)f * FT8RX_NSAMP;
    for (int i = tid; i < 2400; i += 256) {
        int c = i & 7, n1 = i >> 3;
        int m = 320 * n1 + n2b + c;
        float re = 0.0f, im = 0.0f;
        if (2 * m < FT8RX_NSAMP) { short2 v = *reinterpret_cast<const short2*>(a + 2 * m); re = (float)v.x; im = (float)v.y; }
        bufA[c * 300 + n1] = make_float2(re, im);
    }
    __syncthreads();
    cpx* r = lds_fft<300, 5, 5, 4, 3>(bufA, bufB, T.W300, 8, tid, 256);
    cpx* out = A + (size_t)f * 96000;
    for (int i = tid; i < 2400; i += 256) {
        int c = i & 7, k1 = i >> 3, n2 = n2b + c;
        cpx v = r[c * 300 + k1];
        if (n2 * k1 != 0) v = cmul(v, T.W96000[n2 * k1]);
        out[k1 * 320 + n2] = v;
    }
}

__global__ __launch_bounds__(256) void k_cyc_b(const cpx* __restrict__ A, cpx* __restrict__ Z, Tables T) {
    __shared__ cpx bufA[4 * 320];
    __shared__ cpx bufB[4 * 320];
    const int f = blockIdx.y, tid = threadIdx.x, k1b = 4 * blockIdx.x;
    const cpx* in = A + (size_t)f * 96000 + (size_t)k1b * 320;
    for (int i = tid; i < 1280; i += 256) bufA[i] = in[i];
    __syncthreads();
    cpx* r = lds_fft<320, 8, 8, 5>(bufA, bufB, T.W320, 4, tid, 256);
    cpx* out = Z + (size_t)f * 96000;
    for (int i = tid; i < 1280; i += 256) {
        int rr = i & 3, k2 = i >> 2;
        out[(k1b + rr) + 300 * k2] = r[rr * 320 + k2];
    }
}

__global__ __launch_bounds__(256) void k_cyc_c(const cpx* __restrict__ Z, cpx* __restrict__ spec, Tables T) {
    const int f = blockIdx.y, k = blockIdx.x * 256 + threadIdx.x;
    const cpx* z = Z + (size_t)f * 96000;
    cpx p = z[k], q = z[(96000 - k) % 96000];
    float er = 0.5f * (p.x + q.x), ei = 0.5f * (p.y - q.y);
    float orr = 0.5f * (p.y + q.y), oi = 0.5f * (q.x - p.x);
    cpx w = T.WR192k[k];
    spec[(size_t)f * FT8RX_SPEC_BINS + k] = make_float2(er + (w.x * orr - w.y * oi), ei + (w.x * oi + w.y * orr));
}

// ------------------------------------------------------------------------------------ fine time/frequency sync (receiver.py:140-206)
// One 128-thread block per candidate.  The 3200-point inverse FFT (conj o forward o conj) runs in place in
// one LDS buffer as three register-fused stages [8] | [4,4] | [5,5]; stage 1 reads the tapered spectrum
// slice straight from global memory and knows that only 1000 of the 3200 bins are non-zero.  The 1/3200
// scale and the output conjugation are applied where the series is consumed.
#ifndef FINE_NT
#define FINE_NT 128
#endif
#ifndef FINE_WV
#define FINE_WV 2
#endif
#define FINE_INV 0.0003125f

// conj(taper * spec) for bin k of the rolled 3200-bin slice (receiver.py:180-185); k < 850 or k >= 3050.
// `sl` is the candidate's spectrum window staged in LDS: sl[i] = spec[fb0 - 182 + i], i < 1064 (covers every ftweak);
// off = ftweak + 182.
#define FINE_SLICE 1064
FT8_DEV cpx fine_input(const cpx* sl, int off, int k, const double* __restrict__ taper) {
    cpx v; int ti;
    if (k < 850) { v = sl[off + k]; ti = (k >= 750) ? k - 750 : -1; }
    else { const int j = k - 3050; v = sl[off - 150 + j]; ti = (j < 100) ? j : -1; }
    if (ti >= 0) { const double t = taper[ti]; v.x = (float)((double)v.x * t); v.y = (float)((double)v.y * t); }
    v.y = -v.y;
    return v;
}

// forward FFT of the conjugated slice into z (unscaled, unconjugated), natural Stockham layout, in place.
// (Measured alternatives, profiles/r01_notes.md: 256-thread blocks 7.96 ms, bank-conflict-free padded/transposed
// inter-stage layouts 6.92 ms, this version 6.36 ms per 256 frames: the kernel is latency/barrier bound.)

// The stages are written for any FINE_NT in {64, 128}: a thread owns ceil(groups / FINE_NT) groups of each stage,
// loads all of them, passes the barrier, then computes and stores them (in place).  With FINE_NT = 64 the block is a
// single wavefront, the "barriers" are free and every lane carries 3-4 independent groups (ILP instead of TLP).
FT8_DEV void fine_stage1(const cpx* S, int fb, cpx* z, const cpx* __restrict__ W,
                         const double* __restrict__ taper, int tid) {
    // pass [8]: n = 3200, s = 1, m = 400: butterfly p reads bins p + 400 j; only j = 0, 1, (2 if p < 50), (7 if p >= 250)
    // are non-zero.  All global loads of the thread are issued before the first butterfly.
    constexpr int R = (400 + FINE_NT - 1) / FINE_NT;
    const cpx zero = make_float2(0.0f, 0.0f);
    cpx in0[R], in1[R], in7[R], in2;
#pragma unroll
    for (int i = 0; i < R; i++) {
        const int p = tid + FINE_NT * i;
        const bool on = p < 400;
        in0[i] = on ? fine_input(S, fb, p, taper) : zero;
        in1[i] = on ? fine_input(S, fb, p + 400, taper) : zero;
        in7[i] = (on && p >= 250) ? fine_input(S, fb, p + 2800, taper) : zero;
    }
    in2 = (tid < 50) ? fine_input(S, fb, tid + 800, taper) : zero;
#pragma unroll
    for (int i = 0; i < R; i++) {
        const int p = tid + FINE_NT * i;
        if (p < 400) {
            cpx a[8];
            a[0] = in0[i]; a[1] = in1[i]; a[2] = (i == 0) ? in2 : zero;
            a[3] = zero; a[4] = zero; a[5] = zero; a[6] = zero; a[7] = in7[i];
            dft<8>(a);
            z[8 * p] = a[0];
#pragma unroll
            for (int j = 1; j < 8; j++) { cpx v = a[j]; if (p != 0) v = cmul(v, W[j * p]); z[8 * p + j] = v; }
        }
    }
    __syncthreads();
}
FT8_DEV void fine_stage2(cpx* z, const cpx* w400, int tid) {
    typedef Fused2<3200, 400, 8, 4, 4> F;                         // passes [4,4]: n = 400, s = 8; 200 groups
    constexpr int R = (F::groups + FINE_NT - 1) / FINE_NT;
    cpx a[R][4][4];
    // group g = (pp = g / 8, q = g % 8): in  q + 8(pp + 25 j' + 100 j),  out  q + 8 j + 32 (4 pp + j')
#pragma unroll
    for (int r = 0; r < R; r++) { const int g = tid + FINE_NT * r; if (g < F::groups) F::load_affine<200, 800>(z, g, a[r]); }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < R; r++) {
        const int g = tid + FINE_NT * r;
        if (g < F::groups) {
            // same arithmetic as F::compute_pp; every twiddle index of this stage is a multiple of 8, so the factors come
            // from the 400-entry LDS copy w400[t] = W3200[8 t]:  pass A  W3200[j p 8] = w400[j p],  pass B  W3200[j' pp 32] = w400[4 j' pp]
            const int pp = g >> 3;
#pragma unroll
            for (int jp = 0; jp < 4; jp++) {
                dft<4>(a[r][jp]);
                const int pq = pp + 25 * jp;
                if (pq != 0) {
#pragma unroll
                    for (int j = 1; j < 4; j++) a[r][jp][j] = cmul(a[r][jp][j], w400[j * pq]);
                }
            }
#pragma unroll
            for (int j = 0; j < 4; j++) {
                cpx b[4];
#pragma unroll
                for (int jp = 0; jp < 4; jp++) b[jp] = a[r][jp][j];
                dft<4>(b);
                if (pp != 0) {
#pragma unroll
                    for (int jp = 1; jp < 4; jp++) b[jp] = cmul(b[jp], w400[4 * jp * pp]);
                }
#pragma unroll
                for (int jp = 0; jp < 4; jp++) a[r][jp][j] = b[jp];
            }
            F::store_affine<32, 8>(z, (g & 7) + 128 * (g >> 3), a[r]);
        }
    }
    __syncthreads();
}
// Only output samples in [lo, hi) are needed (the scoring IFFTs read one Costas block = ~230 samples): a final
// radix-5 butterfly (q, j) produces samples q + 128 j + 640 j', at most one of which can fall in a window
// shorter than 640, so butterflies with no sample in the window are skipped.  Needed outputs are bit-identical.
FT8_DEV void fine_stage3(cpx* z, const cpx* __restrict__ W, int tid, int lo, int hi) {
    typedef Fused2<3200, 25, 128, 5, 5> F;                        // passes [5,5]: n = 25, s = 128; 128 groups
    constexpr int R = F::groups / FINE_NT;
    cpx a[R][5][5];
    // group q: in  q + 128 (j' + 5 j),  out  q + 128 j + 640 j'
#pragma unroll
    for (int r = 0; r < R; r++) F::load_affine<128, 640>(z, tid + FINE_NT * r, a[r]);
    __syncthreads();
#pragma unroll
    for (int r = 0; r < R; r++) {
        const int q = tid + FINE_NT * r;
        F::compute_passA(0, a[r], W);
#pragma unroll
        for (int j = 0; j < 5; j++) {
            const int rr = q + 128 * j;
            const int first = (lo <= rr) ? rr : rr + 640 * ((lo - rr + 639) / 640);   // smallest rr + 640 j' >= lo
            if (first < hi && first < 3200) {
                cpx b[5];
#pragma unroll
                for (int jp = 0; jp < 5; jp++) b[jp] = a[r][jp][j];
                dft<5>(b);                                         // last pass: no twiddles
#pragma unroll
                for (int jp = 0; jp < 5; jp++) z[rr + 640 * jp] = b[jp];
            }
        }
    }
    __syncthreads();
}
FT8_DEV void fine_fft(const cpx* S, int fb, cpx* z, const cpx* w400, const Tables& T, int tid, int lo, int hi) {
    fine_stage1(S, fb, z, T.W3200, T.taper, tid);
    fine_stage2(z, w400, tid);
    fine_stage3(z, T.W3200, tid, lo, hi);
}

// |32-pt DFT| tones 0..7 of the symbol starting at sample i0, computed by the 4 lanes of a quad
FT8_DEV void fine_sym_quad(const cpx* z, int i0, int n2, int lane, const cpx* w32, float* mag) {
    if (i0 < 0) i0 = 0;
    if (i0 > 3168) i0 = 3168;
    cpx x[8];
#pragma unroll
    for (int n1 = 0; n1 < 8; n1++) { cpx v = z[i0 + 4 * n1 + n2]; x[n1] = make_float2(v.x * FINE_INV, -(v.y * FINE_INV)); }
    sym32_quad(x, n2, lane, w32, mag);
}

__global__ __launch_bounds__(FINE_NT, FINE_WV) void k_fine(const cpx* __restrict__ spec, ft8rx_record* __restrict__ rec,
                                                  const int32_t* __restrict__ ncand, float* __restrict__ llr0, Tables T, ft8rx_config cfg,
                                                  const int32_t* __restrict__ trip, int32_t* __restrict__ t_out /*[n][5]*/,
                                                  float* __restrict__ t_sd, float* __restrict__ t_sgrid) {
    __shared__ cpx z[3200];
    __shared__ cpx slice[FINE_SLICE];  // the candidate's 1064 spectrum bins, read by the first stage of all ten IFFTs
    __shared__ __attribute__((aligned(8))) float mg[640];   // scoring (on, off) sums as fp64, later the [79][8] grid
    __shared__ cpx w400[400];          // W3200[8 t]: every twiddle of the [4,4] stage
    float* p = reinterpret_cast<float*>(slice);      // [464] the slice is dead once the last IFFT has run: reuse it
    float* llr = p + 464;                            // [176]
    float* sq = llr + 176;                           // [176]
    __shared__ float sc[16];
    __shared__ int ish[4];
    __shared__ cpx w32[32];
    const int tid = threadIdx.x, lane = tid & 63;
    int frame, ci = 0, f0, h0;
    if (trip) { frame = trip[3 * blockIdx.x]; f0 = trip[3 * blockIdx.x + 1]; h0 = trip[3 * blockIdx.x + 2]; }
    else {
        frame = blockIdx.x / MAXC; ci = blockIdx.x % MAXC;
        if (ci >= ncand[frame]) return;
        const ft8rx_record& r = rec[(size_t)frame * MAXC + ci];
        if (r.status != FT8RX_ST_ACTIVE) return;
        f0 = r.f0_idx; h0 = r.h0_idx;
    }
    if (tid < 32) w32[tid] = T.W32[tid];
    const int fb0 = 50 * f0;                                      // int(0.5 + fHz*16)
    {
        const cpx* __restrict__ Sg = spec + (size_t)frame * FT8RX_SPEC_BINS + (fb0 - 182);
        for (int i = tid; i < FINE_SLICE; i += FINE_NT) slice[i] = Sg[i];
        for (int i = tid; i < 400; i += FINE_NT) w400[i] = T.W3200[8 * i];
        __syncthreads();
    }
    const cpx* S = slice;
    const int tb0 = 8 * h0 + (h0 < 0 ? 1 : 0);                    // int(0.5 + tsec/0.005) truncates toward zero
    // Score of one Costas block (contract): per symbol a the quad leader forms on_a = |tone costas[a]| and off_a = sum of
    // the other six tones (b ascending) in fp64 from its registers; after ONE barrier every thread combines
    // S1 = sum_a on_a, S2 = sum_a off_a (a ascending) and score = (float)(S1 + w6 S2) -- no serial chain, no broadcast.
    double* dsum = reinterpret_cast<double*>(mg);              // [8][7][2] (on, off); mg is free until the final grid
    // --- time tweaks at ftweak 0: range(-8,8,2) -> 8 x 7 symbols, 4 lanes each
    fine_fft(S, 182, z, w400, T, tid, tb0 - 8 + 32 * 36, tb0 + 6 + 32 * 43);   // the 8 time tweaks of the middle Costas block
#pragma unroll 1
    for (int r = 0; r < (224 + FINE_NT - 1) / FINE_NT; r++) {
        const int task = tid + FINE_NT * r, qd = task >> 2, n2 = task & 3;
        const bool valid = qd < 56;
        const int ti = valid ? qd / 7 : 0, a = valid ? qd - 7 * ti : 0;
        float mag[8];
        fine_sym_quad(z, tb0 - 8 + 2 * ti + 32 * (36 + a), n2, lane, w32, mag);
        if (valid && n2 == 0) {
            const int c = d_COSTAS[a];
            double off = 0.0, on = 0.0;
#pragma unroll
            for (int b = 0; b < 7; b++) { if (b == c) on = (double)mag[b]; else off += (double)mag[b]; }
            dsum[(ti * 7 + a) * 2] = on; dsum[(ti * 7 + a) * 2 + 1] = off;
        }
    }
    __syncthreads();
    int tt = -8; float score_f0 = 0.0f;
    for (int ti = 0; ti < 8; ti++) {                           // every thread: same values, same result
        double s1 = 0.0, s2 = 0.0;
#pragma unroll
        for (int a = 0; a < 7; a++) { s1 += dsum[(ti * 7 + a) * 2]; s2 += dsum[(ti * 7 + a) * 2 + 1]; }
        const float sct = (float)(s1 + W6 * s2);
        if (ti == 0 || sct > score_f0) { score_f0 = sct; tt = -8 + 2 * ti; }     // first maximum (np.argmax)
    }
    // --- frequency tweaks: range(-32,33,8)
    float best = 0.0f; int ft = 0;
#pragma unroll 1
    for (int i = 0; i < 9; i++) {
        const int fcur = -32 + 8 * i;
        float s;
        if (fcur == 0) s = score_f0;             // same series, same offset: identical value
        else {
            fine_fft(S, 182 + fcur, z, w400, T, tid, tb0 + tt + 32 * 36, tb0 + tt + 32 * 43);
            if (tid < 64) {                       // 7 symbols x 4 lanes on wavefront 0
                const int qd = tid >> 2, n2 = tid & 3;
                const bool valid = qd < 7;
                float mag[8];
                fine_sym_quad(z, tb0 + tt + 32 * (36 + (valid ? qd : 0)), n2, lane, w32, mag);
                if (valid && n2 == 0) {
                    const int c = d_COSTAS[qd];
                    double off = 0.0, on = 0.0;
#pragma unroll
                    for (int b = 0; b < 7; b++) { if (b == c) on = (double)mag[b]; else off += (double)mag[b]; }
                    dsum[qd * 2] = on; dsum[qd * 2 + 1] = off;
                }
            }
            __syncthreads();
            double s1 = 0.0, s2 = 0.0;
#pragma unroll
            for (int a = 0; a < 7; a++) { s1 += dsum[a * 2]; s2 += dsum[a * 2 + 1]; }
            s = (float)(s1 + W6 * s2);
        }
        if (i == 0 || s > best) { best = s; ft = fcur; }
    }
    fine_fft(S, 182 + ft, z, w400, T, tid, 0, 3200);   // full series for the 79 x 8 grid
#pragma unroll 1
    for (int r = 0; r < (316 + FINE_NT - 1) / FINE_NT; r++) {                 // full 79 x 8 grid
        const int task = tid + FINE_NT * r, sy = task >> 2, n2 = task & 3;
        const bool valid = sy < 79;
        float mag[8];
        fine_sym_quad(z, tb0 + tt + 32 * (valid ? sy : 0), n2, lane, w32, mag);
        if (valid && n2 == 0) {
#pragma unroll
            for (int b = 0; b < 8; b++) mg[sy * 8 + b] = mag[b];
        }
    }
    __syncthreads();
    // --- Costas gate (receiver.py:164-167)
    bool match = false;
    if (tid < 21) {
        int blk = tid / 7, a = tid - blk * 7;
        const float* q = mg + 8 * (36 * blk + a);
        int am = 0; for (int t = 1; t < 8; t++) if (q[t] > q[am]) am = t;
        match = (am == d_COSTAS[a]);
    }
    if (tid < 64) { int nm = __popcll(__ballot(match)); if (tid == 0) ish[1] = nm; }
    __syncthreads();
    const int nsync = ish[1];
    if (trip && t_sgrid) for (int i = tid; i < 632; i += FINE_NT) t_sgrid[(size_t)blockIdx.x * 632 + i] = mg[i];
    int ret = 1; float sd = 0.0f; int snr = 0;
    if (nsync <= 6) ret = 0;           // block-uniform
    else {
        for (int i = tid; i < 464; i += FINE_NT) p[i] = 20.0f * ft8_log10f(mg[8 * (int)d_PAYSYM[i >> 3] + (i & 7)]);   // receiver.py:170
        __syncthreads();
        llr_from_p(p, llr, sq, tid, tid < 64, &sd, &snr);
        if (tid == 0) { sc[10] = sd; ish[2] = snr; }
        __syncthreads();
        sd = sc[10]; snr = ish[2];
        if (sd <= cfg.llr_sd_min) ret = -1;
        float* out = llr0 + (size_t)blockIdx.x * 174;
        for (int i = tid; i < 174; i += FINE_NT) out[i] = llr[i];
    }
    if (tid == 0) {
        if (trip) { int32_t* o = t_out + 5 * (size_t)blockIdx.x; o[0] = ret; o[1] = tt; o[2] = ft; o[3] = nsync; o[4] = snr; t_sd[blockIdx.x] = sd; }
        else {
            ft8rx_record& r = rec[(size_t)frame * MAXC + ci];
            r.ttweak = (int8_t)tt; r.ftweak = (int8_t)ft; r.nsync = (uint8_t)nsync;
            if (ret == 0) r.status = FT8RX_ST_STOP_COSTAS;
            else { r.fine_sd = sd; r.snr_fine = (int8_t)snr; if (ret < 0) r.status = FT8RX_ST_STOP_FINE_SD; }
        }
    }
}

// ------------------------------------------------------------------------------------ OSD (decoders.py:223-272)
// One wavefront per attempt.  Lane r holds generator row r (and row 64+r for r<27) in registers.
FT8_DEV uint64_t shfl64(uint64_t v, int src) {
    uint32_t lo = __shfl((uint32_t)v, src), hi = __shfl((uint32_t)(v >> 32), src);
    return ((uint64_t)hi << 32) | lo;
}
FT8_DEV uint64_t xor_reduce64(uint64_t v) {
    for (int o = 32; o > 0; o >>= 1) {
        uint32_t lo = __shfl_xor((uint32_t)v, o), hi = __shfl_xor((uint32_t)(v >> 32), o);
        v ^= ((uint64_t)hi << 32) | lo;
    }
    return v;
}

#define OSD_MAXTRIALS 512
// mode 0: pipeline (work = (candidate, slot 0..9)); mode 2: raw vectors
__global__ __launch_bounds__(64) void k_osd(int mode, const float* __restrict__ llr_in, const float* __restrict__ saved,
                                            const Att* __restrict__ attB, ft8rx_record* __restrict__ rec,
                                            const int32_t* __restrict__ ncand, Att* __restrict__ attO,
                                            ft8rx_event* ev, int32_t* evcount, int singles, int doubles) {
    __shared__ float llr[176];
    __shared__ uint64_t skey[256];
    __shared__ uint64_t flip[64][2];
    const int lane = threadIdx.x;
    int frame = 0, ci = 0, slot = 0; size_t vec = blockIdx.x;
    if (mode == 0) {
        slot = blockIdx.x % 10; int c = blockIdx.x / 10; frame = c / MAXC; ci = c % MAXC;
        if (ci >= ncand[frame]) return;
        if (rec[(size_t)frame * MAXC + ci].status != FT8RX_ST_ACTIVE) return;
        if (slot < 5) { for (int i = lane; i < 174; i += 64) llr[i] = ap_value(slot, i, llr_in[(size_t)c * 174 + i]); }
        else {
            if (!attB[(size_t)c * 5 + (slot - 5)].has_out) { if (lane == 0) { Att a; memset(&a, 0, sizeof(a)); a.n_its = -1; attO[(size_t)c * 10 + slot] = a; } return; }
            for (int i = lane; i < 174; i += 64) llr[i] = saved[((size_t)c * 5 + (slot - 5)) * 174 + i];
        }
        vec = (size_t)c * 10 + slot;
    } else {
        for (int i = lane; i < 174; i += 64) llr[i] = llr_in[vec * 174 + i];
    }
    __syncthreads();
    // ---- reliability order: |llr| descending, ties and NaNs (last) by index (fixed rule for np.argsort, decoders.py:226).
    // Bitonic network over 256 composite keys ((~magnitude bits) << 32 | index) in LDS: 36 compare-exchange steps.
    for (int i = lane; i < 256; i += 64) {
        uint64_t key = ~0ull;
        if (i < 174) {
            const float x = llr[i];
            const uint32_t k32 = (x != x) ? 0u : ((__float_as_uint(x) & 0x7fffffffu) + 1u);
            key = ((uint64_t)(0xFFFFFFFFu - k32) << 32) | (uint32_t)i;
        }
        skey[i] = key;
    }
    __syncthreads();
    for (int size = 2; size <= 256; size <<= 1) {
        for (int stride = size >> 1; stride > 0; stride >>= 1) {
#pragma unroll
            for (int h2 = 0; h2 < 2; h2++) {
                const int t = lane + 64 * h2;
                const int pos = ((t & ~(stride - 1)) << 1) | (t & (stride - 1));
                const uint64_t ka = skey[pos], kb = skey[pos + stride];
                const bool up = ((pos & size) == 0);
                if ((ka > kb) == up) { skey[pos] = kb; skey[pos + stride] = ka; }
            }
            __syncthreads();
        }
    }
    // ---- Gauss-Jordan over GF(2), most-reliable-basis selection.  The sorted column order and the hard
    // decisions are lifted into registers / wave-uniform masks so the dependent chain of one elimination step
    // is readlane -> bit test -> ballot -> ctz -> readlane (no LDS access on the critical path).
    const int ord0 = (int)(uint32_t)skey[lane], ord1 = (int)(uint32_t)skey[64 + lane], ord2 = (lane < 46) ? (int)(uint32_t)skey[128 + lane] : 0;
    const uint64_t hard0 = __ballot(llr[lane] > 0.0f), hard1 = __ballot(llr[64 + lane] > 0.0f),
                   hard2 = __ballot(lane < 46 && llr[128 + (lane < 46 ? lane : 0)] > 0.0f);
    uint64_t a0 = d_G0[lane][0], a1 = d_G0[lane][1], a2 = d_G0[lane][2];
    const bool hasB = lane < 27;
    uint64_t b0 = hasB ? d_G0[64 + lane][0] : 0, b1 = hasB ? d_G0[64 + lane][1] : 0, b2 = hasB ? d_G0[64 + lane][2] : 0;
    // Basis exchange.  G0 = [I | A^T] is already reduced for the systematic basis: row r owns unit column r.
    // Columns are visited in reliability order exactly as in the reference (decoders.py:228-242) and accepted
    // iff independent of the columns accepted so far, but
    //   * a row is "locked" once its basis column has been accepted; an UNLOCKED row r always still owns its original
    //     column r (rows only change basis column at the moment they are locked), so "column c is the unit column of
    //     an unlocked row" is the wave-uniform test  c < 91 && !locked(c): such a column is accepted by setting one
    //     bit -- no row operation, no ballot, no broadcast;
    //   * any other column is accepted iff it has a 1 in some unlocked row; one elimination step then makes it that
    //     row's unit column.
    // The selected basis, the reduced rows and the acceptance order k are identical to plain Gauss-Jordan; about 40 %
    // of the accepted columns need no row operation.  All bookkeeping is wave-uniform (scalar registers):
    // lockA/lockB = locked rows 0..63 / 64..90, hmA/hmB = locked rows whose accepted column has hard decision 1.
    uint64_t lockA = 0, lockB = ~((1ull << 27) - 1), hmA = 0, hmB = 0;
    __shared__ uint8_t prow[96];                             // prow[k] = row locked by the k-th accepted column
    int k = 0;
    for (int ic = 0; ic < 174 && k < 91; ic++) {
        const int sel = ic >> 6, il = ic & 63;
        const int col = __builtin_amdgcn_readlane(sel == 0 ? ord0 : (sel == 1 ? ord1 : ord2), il);
        const int w = col >> 6, sh = col & 63;
        const uint64_t hw = (w == 0) ? hard0 : (w == 1) ? hard1 : hard2;
        const uint64_t hard = (hw >> sh) & 1ull;
        int row = -1;
        if (col < 91 && !(((col < 64 ? lockA : lockB) >> (col & 63)) & 1ull)) row = col;      // still a unit column
        else {
            const uint64_t wa = (w == 0) ? a0 : (w == 1) ? a1 : a2;
            const uint64_t wb = (w == 0) ? b0 : (w == 1) ? b1 : b2;
            const bool bitA = (wa >> sh) & 1ull, bitB = (wb >> sh) & 1ull;
            const uint64_t mA = __ballot(bitA) & ~lockA, mB = __ballot(bitB) & ~lockB;
            if (!mA && !mB) continue;                        // dependent on the accepted columns
            const bool inA = (mA != 0);
            const int src = inA ? __builtin_ctzll(mA) : __builtin_ctzll(mB);
            const uint64_t p0 = shfl64(inA ? a0 : b0, src), p1 = shfl64(inA ? a1 : b1, src), p2 = shfl64(inA ? a2 : b2, src);
            if (bitA && !(inA && lane == src)) { a0 ^= p0; a1 ^= p1; a2 ^= p2; }
            if (bitB && !(!inA && lane == src)) { b0 ^= p0; b1 ^= p1; b2 ^= p2; }
            row = inA ? src : 64 + src;
        }
        if (row < 64) { lockA |= 1ull << row; hmA |= hard << row; }
        else { lockB |= 1ull << (row - 64); hmB |= hard << (row - 64); }
        if (lane == 0) prow[k] = (uint8_t)row;
        k++;
    }
    // order-0 codeword (message part = first 91 bits): XOR of the locked rows whose accepted column has hard bit 1
    const bool hardA = (hmA >> lane) & 1ull, hardB = (hmB >> lane) & 1ull;
    uint64_t c0 = (hardA ? a0 : 0) ^ (hardB ? b0 : 0), c1 = (hardA ? a1 : 0) ^ (hardB ? b1 : 0);
    c0 = xor_reduce64(c0); c1 = xor_reduce64(c1);
    __syncthreads();
    // flip rows: flip[i] = row locked by accepted column 90 - i (the least reliable basis members first)
    {
        const int i = lane;
        const int r = (i < 64 && 90 - i >= 0 && 90 - i < k) ? prow[90 - i] : 0;
        const uint64_t fa0 = shfl64(a0, r & 63), fa1 = shfl64(a1, r & 63), fb0 = shfl64(b0, r & 63), fb1 = shfl64(b1, r & 63);
        flip[i][0] = (r < 64) ? fa0 : fb0;
        flip[i][1] = (r < 64) ? fa1 : fb1;
    }
    __syncthreads();
    // trial t in the reference's order (decoders.py:248-272): 0 = order-0, 1..S = single flips i = t-1, then the
    // restricted double flips (i, j), i < S, j < min(i, D), i-major.
    int npairs = 0;
    for (int i = 0; i < singles; i++) npairs += (i < doubles) ? i : doubles;
    const int ntr = 1 + singles + npairs;
    const int dtri = doubles * (doubles - 1) / 2;          // pairs with i < D
    const uint64_t M1 = (1ull << 27) - 1;
    Att res; memset(&res, 0, sizeof(res)); res.n_its = -1;
    const int ipass = (slot < 5) ? 5 : 6;
    for (int base = 0; base < ntr; base += 64) {
        const int t = base + lane;
        int r = 0; uint64_t lo = 0, hi = 0;
        if (t < ntr) {
            int i = -1, j = -1;
            if (t >= 1 && t <= singles) i = t - 1;
            else if (t > singles) {
                const int u = t - 1 - singles;
                if (u < dtri) { i = 1; while ((i + 1) * i / 2 <= u) i++; j = u - i * (i - 1) / 2; }
                else { const int v = u - dtri; i = doubles + v / doubles; j = v - (v / doubles) * doubles; }
            }
            uint64_t w0 = c0, w1 = c1;
            if (i >= 0) { w0 ^= flip[i][0]; w1 ^= flip[i][1]; }
            if (j >= 0) { w0 ^= flip[j][0]; w1 ^= flip[j][1]; }
            r = ft8_crc_check(w0, w1 & M1, &lo, &hi);
        }
        const uint64_t acc = __ballot(r == 2);
        const int win = acc ? __builtin_ctzll(acc) : 64;
        if (r && lane <= win) log_event(ev, evcount, frame, ci, ipass, slot, t, lo, hi, r == 2);   // calls the reference made
        if (acc) {
            res.ok = 1; res.lo = shfl64(lo, win); res.hi = shfl64(hi, win); res.n_its = (int16_t)(base + win);
            res.method = (slot < 5) ? FT8RX_M_OSD : FT8RX_M_LDPC_B_OSD;
            break;
        }
    }
    if (lane == 0) attO[vec] = res;
}

// ------------------------------------------------------------------------------------ synthetic frames
// Device twin of pyft8_amd/synth.py: 79-tone GFSK (BT = 2.0, reference transmitter.py:41-70 as the model) for up to
// 64 signals per frame + unit-variance white noise from a counter-based Philox4x32-10 stream, scaled to sigma = 1000
// counts and clipped to int16.  Workload generator only -- not on the receive path.
struct SynthSig {            // one signal; filled by the host (pyft8_amd/synth.py: device_signal_table)
    double f0;               // Hz
    double cum[82];          // cum[i] = sum_{i'<i} ext[i'] * (Qtot - Qs[i'])   (fully integrated symbols)
    float amp;               // linear amplitude relative to unit-variance noise
    int32_t i0;              // first sample of the 79-symbol waveform inside the frame
    uint8_t ext[84];         // 81 extended tones (first and last repeated), padded
};

FT8_DEV void philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1, uint32_t* out) {
#pragma unroll
    for (int r = 0; r < 10; r++) {
        const uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
        const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0, n1 = (uint32_t)p1, n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1, n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

// one thread = 4 consecutive samples of one frame
__global__ __launch_bounds__(256) void k_synth(int16_t* __restrict__ audio, const SynthSig* __restrict__ sigs, int nsig,
                                               const double* __restrict__ Q /*[5761]*/, uint32_t seed_lo, uint32_t seed_hi, int first_index) {
    const int f = blockIdx.y;
    const int n0 = 4 * (blockIdx.x * 256 + threadIdx.x);
    if (n0 >= FT8RX_NSAMP) return;
    uint32_t r[4];
    philox4x32_10((uint32_t)(n0 >> 2), (uint32_t)(first_index + f), 0u, 0u, seed_lo, seed_hi, r);
    double x[4];
    {   // Box-Muller: 2 uniform pairs -> 4 normals
        const double u0 = ((double)r[0] + 0.5) * (1.0 / 4294967296.0), u1 = ((double)r[1] + 0.5) * (1.0 / 4294967296.0);
        const double u2 = ((double)r[2] + 0.5) * (1.0 / 4294967296.0), u3 = ((double)r[3] + 0.5) * (1.0 / 4294967296.0);
        const double ra = sqrt(-2.0 * log(u0)), rb = sqrt(-2.0 * log(u2));
        x[0] = ra * cos(6.283185307179586 * u1); x[1] = ra * sin(6.283185307179586 * u1);
        x[2] = rb * cos(6.283185307179586 * u3); x[3] = rb * sin(6.283185307179586 * u3);
    }
    const SynthSig* S = sigs + (size_t)f * nsig;
    const double Qtot = Q[5760];
    for (int sg = 0; sg < nsig; sg++) {
        const int i0 = S[sg].i0;
        if (n0 + 3 < i0 || n0 >= i0 + 79 * 1920) continue;
        const double f0 = S[sg].f0; const float amp = S[sg].amp;
        for (int k = 0; k < 4; k++) {
            const int m = n0 + k - i0;
            if (m < 0 || m >= 79 * 1920) continue;
            int ih = (m + 3840) / 1920; if (ih > 80) ih = 80;
            const int il = ih - 2 > 0 ? ih - 2 : 0;
            double acc = S[sg].cum[il];
            for (int i = il; i <= ih; i++) {
                int qi = m + 3840 - 1920 * i; if (qi > 5760) qi = 5760;
                const double qs = (i == 0) ? Q[3840] : (i == 1) ? Q[1920] : 0.0;
                acc += (double)S[sg].ext[i] * (Q[qi] - qs);
            }
            (void)Qtot;
            double phi = 6.283185307179586 * (f0 * (double)m + 6.25 * acc) / 12000.0;
            double w = sin(phi);
            if (m < 240) w *= 0.5 * (1.0 - cos(3.141592653589793 * (double)m / 240.0));
            else if (m >= 79 * 1920 - 240) w *= 0.5 * (1.0 - cos(3.141592653589793 * (double)(79 * 1920 - 1 - m) / 240.0));
            x[k] += (double)amp * w;
        }
    }
    short4 o;
    double v;
    v = rint(x[0] * 1000.0); o.x = (short)(v > 32767.0 ? 32767.0 : (v < -32768.0 ? -32768.0 : v));
    v = rint(x[1] * 1000.0); o.y = (short)(v > 32767.0 ? 32767.0 : (v < -32768.0 ? -32768.0 : v));
    v = rint(x[2] * 1000.0); o.z = (short)(v > 32767.0 ? 32767.0 : (v < -32768.0 ? -32768.0 : v));
    v = rint(x[3] * 1000.0); o.w = (short)(v > 32767.0 ? 32767.0 : (v < -32768.0 ? -32768.0 : v));
    *reinterpret_cast<short4*>(audio + (size_t)f * FT8RX_NSAMP + n0) = o;
}

// ------------------------------------------------------------------------------------ small probes (tests)
__global__ void k_math_probe(int which, const float* x, float* y, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) y[i] = (which == 0) ? ft8_log10f(x[i]) : ft8_tanhf(x[i]);
}
template <int N, int... Rs>
__global__ void k_fft_probe(const cpx* x, cpx* y, const cpx* W) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    cpx* a = reinterpret_cast<cpx*>(smem); cpx* b = a + N;
    for (int i = threadIdx.x; i < N; i += blockDim.x) a[i] = x[i];
    __syncthreads();
    cpx* r = lds_fft<N, Rs...>(a, b, W, 1, threadIdx.x, blockDim.x);
    for (int i = threadIdx.x; i < N; i += blockDim.x) y[i] = r[i];
}
__global__ void k_crc_probe(const float* cw91, int n, int32_t* res, uint64_t* lo, uint64_t* hi) {
    int v = blockIdx.x; int lane = threadIdx.x;
    const float* c = cw91 + (size_t)v * 91;
    uint64_t b0 = __ballot(c[lane] > 0.0f);
    uint64_t b1 = __ballot(lane < 27 && c[64 + (lane < 27 ? lane : 0)] > 0.0f);
    uint64_t l, h; int r = ft8_crc_check(b0, b1, &l, &h);
    if (lane == 0) { res[v] = r; lo[v] = l; hi[v] = h; }
}
__global__ void k_valid_probe(const uint64_t* lo, const uint64_t* hi, int n, int32_t* out) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = ft8_valid77(lo[i], hi[i]) ? 1 : 0;
}

// ====================================================================================== host message layer (native)
// C++ twin of pyft8_amd/messages.py: 77-bit payload -> text (reference decoders.py:16-115), call-hash table
// (databases.py:8-26) and the per-frame replay of records/events in the reference's emit order with its duplicate
// filter (receiver.py:51-66, 389-398).  Pure host code, no HIP: frames are independent and are packaged by a pool
// of threads so that the Python surface is not the bottleneck behind ~26 k decoded frames/s.
#include <algorithm>
#include <thread>
#include <unordered_map>
namespace hostmsg {
static const char A37[] = " 0123456789ABCDEFGHIJKLMNOPQRSTUVWXYZ";
static const char A38[] = " 0123456789ABCDEFGHIJKLMNOPQRSTUVWXYZ/";
static const char A27[] = " ABCDEFGHIJKLMNOPQRSTUVWXYZ";
struct Hashes {
    std::unordered_map<uint64_t, std::string> m;                 // key = nbits << 32 | hash
    void add(const std::string& call) {
        uint64_t acc = 0;
        for (int i = 0; i < 11; i++) {
            char ch = i < (int)call.size() ? call[i] : ' ';
            const char* q = strchr(A38, ch);
            int64_t idx = (q && ch) ? (int64_t)(q - A38) : -1;
            acc = acc * 38 + (uint64_t)idx;
        }
        acc *= 47055833459ULL;
        const int nb[3] = {10, 12, 22};
        for (int k = 0; k < 3; k++) m[((uint64_t)nb[k] << 32) | (acc >> (64 - nb[k]))] = call;
    }
    std::string get(uint32_t h, int nb) const { auto it = m.find(((uint64_t)nb << 32) | h); return it == m.end() ? std::string("...") : it->second; }
};
static std::string strip(const std::string& t) {
    size_t a = 0, b = t.size();
    while (a < b && t[a] == ' ') a++;
    while (b > a && t[b - 1] == ' ') b--;
    return t.substr(a, b - a);
}
static bool plausible(const std::string& c) {
    if (c.size() < 3 || c.find(' ') != std::string::npos) return false;
    auto dig = [](char x) { return x >= '0' && x <= '9'; };
    auto a36 = [](char x) { return (x >= '0' && x <= '9') ? x - '0' : (x >= 'A' && x <= 'Z') ? x - 'A' + 10 : -1; };
    if (c[0] >= 'A' && c[0] <= 'Z' && ((FT8_PFX1_MASK >> (c[0] - 'A')) & 1u) && dig(c[1]))
        if (!(((FT8_PFX1_TRAP >> (c[0] - 'A')) & 1u) && dig(c[2]))) return true;
    int x0 = a36(c[0]), x1 = a36(c[1]);
    return x0 >= 0 && x1 >= 0 && ((FT8_PFX2[x0] >> x1) & 1ULL) && dig(c[2]);
}
static bool field29(uint32_t v29, int i3, Hashes& H, std::string& out) {
    const uint32_t flag = v29 & 1u, n28 = v29 >> 1;
    char buf[24];
    if (n28 < 3) { out = n28 == 0 ? "DE" : n28 == 1 ? "QRZ" : "CQ"; return true; }
    if (n28 < 1004) { snprintf(buf, sizeof buf, "CQ %03u", n28 - 3); out = buf; return true; }
    if (n28 < 21443) {
        uint32_t v = n28 - 1003; std::string t(4, ' ');
        for (int i = 3; i >= 0; i--) { t[i] = A27[v % 27]; v /= 27; }
        out = "CQ " + strip(t); return true;
    }
    if (n28 < 2063592u + 4194303u) { out = "<" + H.get(n28 - 2063592u, 22) + ">"; return true; }
    std::string call;
    int64_t v = (int64_t)n28 - (2063592 + 4194304);
    if (v < 0) call = "ZZ9ZZZ";                                   // negative-index artefact of the reference at n28 = 6257895
    else {
        char ch[7]; ch[6] = 0;
        ch[5] = A27[v % 27]; v /= 27; ch[4] = A27[v % 27]; v /= 27; ch[3] = A27[v % 27]; v /= 27;
        ch[2] = (char)('0' + v % 10); v /= 10; ch[1] = A37[1 + v % 36]; v /= 36; ch[0] = A37[v % 37];
        call = strip(ch);
    }
    if (!plausible(call)) return false;
    if (flag) {
        call += (i3 == 2) ? "/P" : "/R";
        if (i3 != 2 && !(call[0] == 'A' || call[0] == 'K' || call[0] == 'N' || call[0] == 'W')) return false;
    }
    H.add(call);
    out = call;
    return true;
}
// unpack(): true + 3 fields when the reference returns a tuple; mutates H exactly like the reference
static bool unpack(uint64_t lo, uint64_t hi, Hashes& H, std::string f[3]) {
    if (!lo && !hi) return false;
    const unsigned i3 = (unsigned)(lo & 7u);
    if (i3 == 1 || i3 == 2) {
        const uint32_t g16 = (uint32_t)((lo >> 3) & 0xFFFFu), cb = (uint32_t)((lo >> 19) & 0x1FFFFFFFu);
        const uint32_t ca = (uint32_t)(((lo >> 48) | (hi << 16)) & 0x1FFFFFFFu), g15 = g16 & 0x7FFFu;
        if (g15 == 0) return false;
        char g[16];
        if (g15 < 32400) { unsigned q = g15 / 1800, r = g15 % 1800; snprintf(g, sizeof g, "%c%c%02u", 'A' + q, 'A' + r / 100, r % 100); }
        else if (g15 <= 32404) { static const char* T5[5] = {"", "", "RRR", "RR73", "73"}; snprintf(g, sizeof g, "%s", T5[g15 - 32400]); }
        else snprintf(g, sizeof g, "%s%+03d", (g16 >> 15) ? "R" : "", (int)g15 - 32435);
        const bool oka = field29(ca, (int)i3, H, f[0]);
        const bool okb = field29(cb, (int)i3, H, f[1]);
        f[2] = g;
        return oka && okb && g[0] != 0;
    }
    if (i3 == 4) {
        const unsigned cq = (unsigned)((lo >> 3) & 1u), rrr = (unsigned)((lo >> 4) & 3u), swp = (unsigned)((lo >> 6) & 1u);
        uint64_t n58 = ((lo >> 7) | (hi << 57)) & ((1ULL << 58) - 1);
        const uint32_t h12 = (uint32_t)((hi >> 1) & 0xFFFu);
        if ((cq != 0) == (rrr != 0)) return false;
        std::string first = cq ? std::string("CQ") : "<" + H.get(h12, 12) + ">";
        std::string t(12, ' ');
        for (int i = 11; i >= 0; i--) { t[i] = A38[n58 % 38]; n58 /= 38; }
        t = strip(t);
        H.add(t);
        static const char* R4[4] = {"", "RRR", "RR73", "73"};
        f[0] = swp ? t : first; f[1] = swp ? first : t; f[2] = R4[rrr];
        return true;
    }
    return false;
}
struct Ev { int cand, ipass, slot, seq; uint64_t lo, hi; };
static int package_frame(const ft8rx_record* rec, int n, const ft8rx_event* ev, int nev, ft8rx_message* out, int cap) {
    std::vector<Ev> E; E.reserve((size_t)nev);
    for (int i = 0; i < nev; i++) E.push_back({ev[i].cand, ev[i].ipass, ev[i].slot, ev[i].seq, ev[i].msg_lo, ev[i].msg_hi});
    std::sort(E.begin(), E.end(), [](const Ev& a, const Ev& b) {
        if (a.cand != b.cand) return a.cand < b.cand; if (a.ipass != b.ipass) return a.ipass < b.ipass;
        if (a.slot != b.slot) return a.slot < b.slot; return a.seq < b.seq; });
    std::vector<int> last(n), order; order.reserve(n);
    for (int i = 0; i < n; i++) {
        const int st = rec[i].status;
        last[i] = st == FT8RX_ST_DECODED ? rec[i].ipass : st == FT8RX_ST_STOP_GRID_SD ? 0 : (st == FT8RX_ST_STOP_COSTAS || st == FT8RX_ST_STOP_FINE_SD) ? 1 : 7;
    }
    Hashes H; std::vector<std::string> seen; int nm = 0;
    for (int rnd = 0; rnd < 8; rnd++) {
        order.clear();
        for (int i = 0; i < n; i++) if (last[i] >= rnd) order.push_back(i);
        if (rnd == 1) std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return rec[a].grid_sd > rec[b].grid_sd; });
        else if (rnd >= 2) std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return rec[a].fine_sd > rec[b].fine_sd; });
        for (int i : order) {
            const ft8rx_record& r = rec[i];
            const bool here = r.status == FT8RX_ST_DECODED && r.ipass == rnd;
            int sslot = -1, sseq = -1;
            if (here) {
                const int m = r.method;
                sslot = r.ap + (m == FT8RX_M_LDPC_B_OSD ? 5 : 0);
                sseq = m == FT8RX_M_GOOD91 ? 0 : (m == FT8RX_M_LDPC_A || m == FT8RX_M_LDPC_B) ? r.n_its + 1 : r.n_its;
            }
            std::string f[3], got[3]; bool have = false;
            Ev key{i, rnd, -1, -1, 0, 0};
            auto it = std::lower_bound(E.begin(), E.end(), key, [](const Ev& a, const Ev& b) {
                if (a.cand != b.cand) return a.cand < b.cand; if (a.ipass != b.ipass) return a.ipass < b.ipass;
                if (a.slot != b.slot) return a.slot < b.slot; return a.seq < b.seq; });
            int pslot = -2, pseq = -2;
            for (; it != E.end() && it->cand == i && it->ipass == rnd; ++it) {
                if (here && (it->slot > sslot || (it->slot == sslot && it->seq > sseq))) break;
                if (it->slot == pslot && it->seq == pseq) continue;          // the same call logged twice
                pslot = it->slot; pseq = it->seq;
                const bool ok = unpack(it->lo, it->hi, H, f);
                if (here && it->slot == sslot && it->seq == sseq) { have = ok; if (ok) { got[0] = f[0]; got[1] = f[1]; got[2] = f[2]; } }
            }
            if (!here) continue;
            if (!have) { have = unpack(r.msg_lo, r.msg_hi, H, got); if (!have) continue; }    // event log truncated
            std::string text = got[0] + " " + got[1] + " " + got[2];
            if (std::find(seen.begin(), seen.end(), text) != seen.end()) continue;
            seen.push_back(text);
            if (nm < cap) {
                ft8rx_message& o = out[nm]; memset(&o, 0, sizeof(o));
                snprintf(o.f[0], 16, "%s", got[0].c_str()); snprintf(o.f[1], 16, "%s", got[1].c_str()); snprintf(o.f[2], 16, "%s", got[2].c_str());
                o.cand = (int16_t)i; o.f0_idx = r.f0_idx; o.h0_idx = r.h0_idx; o.ipass = r.ipass; o.ap = r.ap; o.method = r.method;
                const bool fine = rnd >= 2;
                o.fine = fine; o.snr = fine ? r.snr_fine : r.snr_grid; o.ttweak = fine ? r.ttweak : 0; o.ftweak = fine ? r.ftweak : 0;
            }
            nm++;
        }
    }
    return nm;
}
}  // namespace hostmsg

// ====================================================================================== host side
#define HIPCHK(h, x) do { hipError_t _e = (x); if (_e != hipSuccess) { set_err(h, "%s failed: %s (%s:%d)", #x, hipGetErrorString(_e), __FILE__, __LINE__); return -2; } } while (0)

static size_t sync_lds_bytes(const ft8rx_config& c) { const size_t nrows = (size_t)(c.h0_hi - c.h0_lo + 24); return nrows * 16 * sizeof(double) + (nrows * 29 + 512) * sizeof(float); }
static std::string g_create_err;

struct ft8rx_handle {
    ft8rx_config cfg;
    int device, max_frames;
    hipStream_t stream;
    int n_streams;                       // chunks of a batch run their kernel chains on separate streams
    hipStream_t sub[8];
    hipEvent_t ev_fork, ev_join[8];
    Tables T;
    std::vector<void*> allocs;
    int16_t* d_audio;            // staging for host-pointer entry points
    float* d_grid;
    float* d_best_score; int32_t* d_best_h0;
    ft8rx_record* d_rec; int32_t* d_ncand;
    float* d_llr0; float* d_saved;
    Att *d_att0, *d_attG, *d_attB, *d_attO;
    cpx *d_A, *d_Z, *d_spec;
    ft8rx_event* d_ev; int32_t* d_evcount;
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

template <typename T> static int dalloc(ft8rx_handle* h, T** p, size_t n) {
    void* q = nullptr;
    hipError_t e = hipMalloc(&q, n * sizeof(T));
    if (e != hipSuccess) { set_err(h, "hipMalloc(%zu bytes) failed: %s", n * sizeof(T), hipGetErrorString(e)); return -2; }
    h->allocs.push_back(q); *p = (T*)q; return 0;
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

struct Scratch {     // RAII device scratch for the test entry points
    ft8rx_handle* h; std::vector<void*> p;
    ~Scratch() { for (void* q : p) hipFree(q); }
    template <typename T> T* get(size_t n) { void* q = nullptr; if (hipMalloc(&q, (n ? n : 1) * sizeof(T)) != hipSuccess) return nullptr; p.push_back(q); return (T*)q; }
    template <typename T> T* put(const T* src, size_t n) { T* q = get<T>(n); if (q && hipMemcpy(q, src, n * sizeof(T), hipMemcpyHostToDevice) != hipSuccess) return nullptr; return q; }
};

extern "C" {

int ft8rx_default_config(ft8rx_config* c) {
    if (!c) return -1;
    c->sync_score_min = 85.0f; c->max_cands = 200; c->f0_lo = 32; c->f0_hi = 960; c->h0_lo = -37; c->h0_hi = 87;
    c->bp_nc0_a = 35; c->bp_iters_a = 5; c->bp_nc0_b = 90; c->bp_iters_b = 20; c->osd_single = 30; c->osd_double = 2;
    c->llr_sd_min = 5.0f;
    return 0;
}

int ft8rx_device_count(void) { int n = 0; if (hipGetDeviceCount(&n) != hipSuccess) return 0; return n; }

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
    for (void* p : h->allocs) hipFree(p);
    for (auto e : h->pev) hipEventDestroy(e);
    for (int i = 0; i < 8; i++) { if (h->sub[i]) hipStreamDestroy(h->sub[i]); if (h->ev_join[i]) hipEventDestroy(h->ev_join[i]); }
    if (h->ev_fork) hipEventDestroy(h->ev_fork);
    if (h->stream) hipStreamDestroy(h->stream);
    delete h;
}

int ft8rx_create(const ft8rx_config* cfg, int device, int max_frames, ft8rx_handle** out) {
    if (!cfg || !out || max_frames < 1) { set_err(nullptr, "ft8rx_create: bad arguments"); return -1; }
    if (cfg->max_cands < 1 || cfg->max_cands > MAXC || cfg->f0_lo < 4 || cfg->f0_hi > 960 || cfg->f0_lo >= cfg->f0_hi ||
        cfg->h0_hi <= cfg->h0_lo || cfg->h0_hi - cfg->h0_lo > 352 || cfg->bp_nc0_a > cfg->bp_nc0_b || cfg->bp_iters_a > cfg->bp_iters_b ||
        cfg->osd_single < 0 || cfg->osd_single > 64 || cfg->osd_double < 0 ||
        1 + cfg->osd_single + cfg->osd_single * cfg->osd_double > OSD_MAXTRIALS) {
        set_err(nullptr, "ft8rx_create: configuration out of the supported range"); return -1; }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1) { set_err(nullptr, "ft8rx_create: no HIP device available (this library has no CPU fallback)"); return -3; }
    if (device < 0 || device >= ndev) { set_err(nullptr, "ft8rx_create: device %d out of range (%d devices)", device, ndev); return -1; }
    ft8rx_handle* h = new ft8rx_handle();
    h->cfg = *cfg; h->device = device; h->max_frames = max_frames; h->stream = nullptr; h->profiling = false; h->n_stage = 0;
    h->n_streams = 4; h->ev_fork = nullptr; for (int i = 0; i < 8; i++) { h->sub[i] = nullptr; h->ev_join[i] = nullptr; }
    if (hipSetDevice(device) != hipSuccess || hipStreamCreate(&h->stream) != hipSuccess) { set_err(nullptr, "ft8rx_create: cannot open device %d", device); delete h; return -2; }
    const size_t B = (size_t)max_frames;
    int rc = 0;
    rc |= dalloc(h, &h->d_audio, B * FT8RX_NSAMP);
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
    rc |= dalloc(h, &h->d_A, B * 96000);
    rc |= dalloc(h, &h->d_Z, B * 96000);
    rc |= dalloc(h, &h->d_spec, B * FT8RX_SPEC_BINS);
    rc |= dalloc(h, &h->d_ev, B * FT8RX_EVENT_CAP);
    rc |= dalloc(h, &h->d_evcount, B);
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
    host_twiddle(3840, 976, w);    rc |= upload(h, &h->T.WR3840, w);
    host_twiddle(3200, 3200, w);   rc |= upload(h, &h->T.W3200, w);
    host_twiddle(96000, 96000, w); rc |= upload(h, &h->T.W96000, w);
    host_twiddle(300, 300, w);     rc |= upload(h, &h->T.W300, w);
    host_twiddle(320, 320, w);     rc |= upload(h, &h->T.W320, w);
    host_twiddle(192000, FT8RX_SPEC_BINS, w); rc |= upload(h, &h->T.WR192k, w);
    host_twiddle(32, 32, w);       rc |= upload(h, &h->T.W32, w);
    if (rc) { g_create_err = h->err; ft8rx_destroy(h); return -2; }
    // LDPC tables
    bool ok = true;
    ok &= hipMemcpyToSymbol(HIP_SYMBOL(d_CHK_N), FT8_CHK_N, sizeof(FT8_CHK_N)) == hipSuccess;
    ok &= hipMemcpyToSymbol(HIP_SYMBOL(d_CHK_V), FT8_CHK_V, sizeof(FT8_CHK_V)) == hipSuccess;
    ok &= hipMemcpyToSymbol(HIP_SYMBOL(d_CHK_E0), FT8_CHK_E0, sizeof(FT8_CHK_E0)) == hipSuccess;
    ok &= hipMemcpyToSymbol(HIP_SYMBOL(d_EDGE_V), FT8_EDGE_V, sizeof(FT8_EDGE_V)) == hipSuccess;
    ok &= hipMemcpyToSymbol(HIP_SYMBOL(d_EDGE_C), FT8_EDGE_C, sizeof(FT8_EDGE_C)) == hipSuccess;
    ok &= hipMemcpyToSymbol(HIP_SYMBOL(d_VAR_E), FT8_VAR_E, sizeof(FT8_VAR_E)) == hipSuccess;
    ok &= hipMemcpyToSymbol(HIP_SYMBOL(d_G0), FT8_G0, sizeof(FT8_G0)) == hipSuccess;
    {   // per-check membership masks (3 x 64 bits), so a lane gets its two checks' masks with 6 coalesced loads
        static uint64_t cm[128][3];
        memset(cm, 0, sizeof(cm));
        for (int c = 0; c < 83; c++) for (int j = 0; j < FT8_CHK_N[c]; j++) { int v = FT8_CHK_V[c][j]; cm[c][v >> 6] |= 1ull << (v & 63); }
        ok &= hipMemcpyToSymbol(HIP_SYMBOL(d_CHK_MASK), cm, sizeof(cm)) == hipSuccess;
    }
    {   // CRC-14 syndromes of the 77 unit messages (bit-serial definition, decoders.py:123-129)
        uint16_t syn[77];
        for (int pos = 0; pos < 77; pos++) syn[pos] = (uint16_t)ft8_crc14_serial_host(pos < 64 ? (1ull << pos) : 0ull, pos >= 64 ? (1ull << (pos - 64)) : 0ull);
        ok &= hipMemcpyToSymbol(HIP_SYMBOL(d_CRC_SYN), syn, sizeof(syn)) == hipSuccess;
    }
    if (!ok) { set_err(nullptr, "ft8rx_create: device table upload failed"); ft8rx_destroy(h); return -2; }
    if (sync_lds_bytes(*cfg) > 65536) ok &= hipFuncSetAttribute((const void*)k_sync, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sync_lds_bytes(*cfg)) == hipSuccess;
    if (!ok) { set_err(nullptr, "ft8rx_create: cannot reserve LDS for k_sync"); ft8rx_destroy(h); return -2; }
    // the never-written grid row 0 (receiver.py:240)
    int nfill = (int)B * FT8RX_GRID_COLS;
    k_fill_row0<<<(nfill + 255) / 256, 256, 0, h->stream>>>(h->d_grid, (int)B);
    if (hipStreamSynchronize(h->stream) != hipSuccess) { set_err(nullptr, "ft8rx_create: init kernel failed"); ft8rx_destroy(h); return -2; }
    for (int i = 0; i < 24; i++) { hipEvent_t e; hipEventCreate(&e); h->pev.push_back(e); }
    for (int i = 0; i < 8; i++) { hipStreamCreateWithFlags(&h->sub[i], hipStreamNonBlocking); hipEventCreateWithFlags(&h->ev_join[i], hipEventDisableTiming); }
    hipEventCreateWithFlags(&h->ev_fork, hipEventDisableTiming);
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
static void enqueue_chain(ft8rx_handle* h, const int16_t* d_audio, int f0, int B, hipStream_t s, bool prof) {
    const ft8rx_config& c = h->cfg;
    const size_t F = (size_t)f0;
    const int16_t* audio = d_audio + F * FT8RX_NSAMP;
    float* grid = h->d_grid + F * FT8RX_GRID_ROWS * FT8RX_GRID_COLS;
    float* bs = h->d_best_score + F * NF0MAX; int32_t* bh = h->d_best_h0 + F * NF0MAX;
    ft8rx_record* rec = h->d_rec + F * MAXC; int32_t* ncand = h->d_ncand + F;
    float* llr0 = h->d_llr0 + F * MAXC * 174; float* saved = h->d_saved + F * MAXC * 5 * 174;
    Att* att0 = h->d_att0 + F * MAXC * 5; Att* attG = h->d_attG + F * MAXC * 2; Att* attB = h->d_attB + F * MAXC * 5; Att* attO = h->d_attO + F * MAXC * 10;
    cpx* A = h->d_A + F * 96000; cpx* Z = h->d_Z + F * 96000; cpx* spec = h->d_spec + F * FT8RX_SPEC_BINS;
    ft8rx_event* ev = h->d_ev + F * FT8RX_EVENT_CAP; int32_t* evc = h->d_evcount + F;
#define STAGE(name) do { if (prof) { hipEventRecord(h->pev[h->pnames.size()], s); h->pnames.push_back(name); } } while (0)
    hipMemsetAsync(evc, 0, sizeof(int32_t) * B, s);
    STAGE("spectrogram");
    k_spectrogram<<<dim3(376, B), SPEC_NT, 0, s>>>(audio, grid, h->T);
    STAGE("sync");
    const int ntile = (c.f0_hi - c.f0_lo + 15) / 16;
    k_sync<<<dim3(ntile, B), 256, sync_lds_bytes(c), s>>>(grid, bs, bh, c);
    STAGE("topk");
    k_topk<<<B, 1024, 0, s>>>(bs, bh, rec, ncand, c);
    STAGE("grid_llr");
    k_grid_llr<<<B * MAXC, 64, 0, s>>>(grid, rec, ncand, llr0, c, nullptr, nullptr, nullptr);
    STAGE("bp_grid");
    k_bp<<<B * MAXC * 5, 64, 0, s>>>(0, llr0, rec, ncand, nullptr, att0, nullptr, ev, evc, c, c.bp_nc0_a, c.bp_iters_a);
    STAGE("select0");
    k_select0<<<(B * MAXC + 255) / 256, 256, 0, s>>>(rec, ncand, att0, B);
    STAGE("cycle_fft");
    k_cyc_a<<<dim3(40, B), 256, 0, s>>>(audio, A, h->T);
    k_cyc_b<<<dim3(75, B), 256, 0, s>>>(A, Z, h->T);
    k_cyc_c<<<dim3(FT8RX_SPEC_BINS / 256, B), 256, 0, s>>>(Z, spec, h->T);
    STAGE("fine");
    k_fine<<<B * MAXC, FINE_NT, 0, s>>>(spec, rec, ncand, llr0, h->T, c, nullptr, nullptr, nullptr, nullptr);
    STAGE("bp_fine");
    k_bp<<<B * MAXC * 5, 64, 0, s>>>(1, llr0, rec, ncand, attG, attB, saved, ev, evc, c, c.bp_nc0_b, c.bp_iters_b);
    STAGE("select1");
    k_select1<<<(B * MAXC + 255) / 256, 256, 0, s>>>(rec, ncand, attG, attB, B, c);
    STAGE("osd");
    k_osd<<<B * MAXC * 10, 64, 0, s>>>(0, llr0, saved, attB, rec, ncand, attO, ev, evc, c.osd_single, c.osd_double);
    STAGE("select2");
    k_select2<<<(B * MAXC + 255) / 256, 256, 0, s>>>(rec, ncand, attO, B);
    if (prof) hipEventRecord(h->pev[h->pnames.size()], s);
#undef STAGE
}

int ft8rx_enqueue_batch(ft8rx_handle* h, const int16_t* d_audio, int B) {
    if (!h || !d_audio) return -1;
    if (B < 1 || B > h->max_frames) { set_err(h, "ft8rx_enqueue_batch: n_frames %d outside [1, %d]", B, h->max_frames); return -1; }
    HIPCHK(h, hipSetDevice(h->device));
    h->pnames.clear();
    // profiling mode: one chain on the main stream so that per-stage events bracket whole-batch launches.
    // normal mode: the batch is cut into n_streams chunks whose chains overlap on separate streams -- the ladder
    // kernels (BP, OSD, fine sync) are latency bound, so chunks fill each other's stalls.
    int ns = h->profiling ? 1 : h->n_streams;
    if (ns > B / 8) ns = B / 8;
    if (ns <= 1) { enqueue_chain(h, d_audio, 0, B, h->stream, h->profiling); HIPCHK(h, hipGetLastError()); return 0; }
    HIPCHK(h, hipEventRecord(h->ev_fork, h->stream));
    const int per = (B + ns - 1) / ns;
    for (int i = 0; i < ns; i++) {
        const int f0 = i * per, n = (f0 + per <= B) ? per : B - f0;
        if (n <= 0) break;
        HIPCHK(h, hipStreamWaitEvent(h->sub[i], h->ev_fork, 0));
        enqueue_chain(h, d_audio, f0, n, h->sub[i], false);
        HIPCHK(h, hipEventRecord(h->ev_join[i], h->sub[i]));
        HIPCHK(h, hipStreamWaitEvent(h->stream, h->ev_join[i], 0));
    }
    HIPCHK(h, hipGetLastError());
    return 0;
}

int ft8rx_set_streams(ft8rx_handle* h, int n) { if (!h || n < 1 || n > 8) return -1; h->n_streams = n; return 0; }

int ft8rx_sync(ft8rx_handle* h) {
    if (!h) return -1;
    HIPCHK(h, hipSetDevice(h->device));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    if (h->profiling && !h->pnames.empty()) {
        h->n_stage = (int)h->pnames.size();
        for (int i = 0; i < h->n_stage; i++) hipEventElapsedTime(&h->stage_ms[i], h->pev[i], h->pev[i + 1]);
    }
    return 0;
}

int ft8rx_fetch_results(ft8rx_handle* h, int B, ft8rx_record* records, int32_t* counts, ft8rx_event* events, int32_t* event_counts) {
    if (!h || B < 1 || B > h->max_frames) return -1;
    HIPCHK(h, hipSetDevice(h->device));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    const int mc = h->cfg.max_cands;
    if (counts) HIPCHK(h, hipMemcpy(counts, h->d_ncand, sizeof(int32_t) * B, hipMemcpyDeviceToHost));
    if (records) HIPCHK(h, hipMemcpy2D(records, sizeof(ft8rx_record) * mc, h->d_rec, sizeof(ft8rx_record) * MAXC, sizeof(ft8rx_record) * mc, B, hipMemcpyDeviceToHost));
    if (event_counts) HIPCHK(h, hipMemcpy(event_counts, h->d_evcount, sizeof(int32_t) * B, hipMemcpyDeviceToHost));
    if (events) HIPCHK(h, hipMemcpy(events, h->d_ev, sizeof(ft8rx_event) * (size_t)B * FT8RX_EVENT_CAP, hipMemcpyDeviceToHost));
    return 0;
}

int ft8rx_decode_batch(ft8rx_handle* h, const int16_t* audio, int B, ft8rx_record* records, int32_t* counts,
                       ft8rx_event* events, int32_t* event_counts) {
    if (!h || !audio) return -1;
    if (B < 1 || B > h->max_frames) { set_err(h, "ft8rx_decode_batch: n_frames %d outside [1, %d]", B, h->max_frames); return -1; }
    HIPCHK(h, hipSetDevice(h->device));
    // Host audio: the batch is cut into chunks (twice the stream count, >= 8 frames each); chunk k's host-to-device copy is
    // issued on its stream right before its kernel chain, so it overlaps the kernels of the chunks before it.
    int nc = h->profiling ? 1 : 2 * h->n_streams;
    if (nc > B / 8) nc = B / 8;
    if (nc <= 1) {
        HIPCHK(h, hipMemcpyAsync(h->d_audio, audio, sizeof(int16_t) * (size_t)B * FT8RX_NSAMP, hipMemcpyHostToDevice, h->stream));
        int rc = ft8rx_enqueue_batch(h, h->d_audio, B);
        if (rc) return rc;
        return ft8rx_fetch_results(h, B, records, counts, events, event_counts);
    }
    h->pnames.clear();
    HIPCHK(h, hipEventRecord(h->ev_fork, h->stream));
    for (int i = 0; i < h->n_streams; i++) HIPCHK(h, hipStreamWaitEvent(h->sub[i], h->ev_fork, 0));
    const int per = (B + nc - 1) / nc;
    for (int k = 0; k < nc; k++) {
        const int f0 = k * per, n = (f0 + per <= B) ? per : B - f0;
        if (n <= 0) break;
        hipStream_t s = h->sub[k % h->n_streams];
        HIPCHK(h, hipMemcpyAsync(h->d_audio + (size_t)f0 * FT8RX_NSAMP, audio + (size_t)f0 * FT8RX_NSAMP,
                                 sizeof(int16_t) * (size_t)n * FT8RX_NSAMP, hipMemcpyHostToDevice, s));
        enqueue_chain(h, h->d_audio, f0, n, s, false);
    }
    for (int i = 0; i < h->n_streams; i++) {
        HIPCHK(h, hipEventRecord(h->ev_join[i], h->sub[i]));
        HIPCHK(h, hipStreamWaitEvent(h->stream, h->ev_join[i], 0));
    }
    HIPCHK(h, hipGetLastError());
    return ft8rx_fetch_results(h, B, records, counts, events, event_counts);
}

// ---------------------------------------------------------------------------- stage entry points
#define NEED(p) do { if (!(p)) { set_err(h, "scratch allocation/copy failed (%s:%d)", __FILE__, __LINE__); return -2; } } while (0)

int ft8rx_spectrogram(ft8rx_handle* h, const int16_t* audio, int B, float* grid) {
    if (!h || !audio || !grid || B < 1 || B > h->max_frames) return -1;
    HIPCHK(h, hipSetDevice(h->device));
    HIPCHK(h, hipMemcpy(h->d_audio, audio, sizeof(int16_t) * (size_t)B * FT8RX_NSAMP, hipMemcpyHostToDevice));
    k_spectrogram<<<dim3(376, B), SPEC_NT, 0, h->stream>>>(h->d_audio, h->d_grid, h->T);
    HIPCHK(h, hipStreamSynchronize(h->stream));
    HIPCHK(h, hipMemcpy(grid, h->d_grid, sizeof(float) * (size_t)B * FT8RX_GRID_ROWS * FT8RX_GRID_COLS, hipMemcpyDeviceToHost));
    return 0;
}

int ft8rx_hop_spectrum(ft8rx_handle* h, const int16_t* window3840, float* row976) {
    if (!h || !window3840 || !row976) return -1;
    HIPCHK(h, hipSetDevice(h->device));
    float* d_row = h->d_best_score;                      // any 976-float scratch: not in use between batches
    HIPCHK(h, hipMemcpyAsync(h->d_audio, window3840, sizeof(int16_t) * 3840, hipMemcpyHostToDevice, h->stream));
    k_hop_spectrum<<<1, SPEC_NT, 0, h->stream>>>(h->d_audio, d_row, h->T);
    HIPCHK(h, hipMemcpyAsync(row976, d_row, sizeof(float) * FT8RX_GRID_COLS, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    return 0;
}

int ft8rx_sync_search(ft8rx_handle* h, const float* grid, int B, int32_t* f0_idx, int32_t* h0_idx, float* score, int32_t* counts) {
    if (!h || !grid || B < 1 || B > h->max_frames) return -1;
    HIPCHK(h, hipSetDevice(h->device));
    const ft8rx_config& c = h->cfg;
    HIPCHK(h, hipMemcpy(h->d_grid, grid, sizeof(float) * (size_t)B * FT8RX_GRID_ROWS * FT8RX_GRID_COLS, hipMemcpyHostToDevice));
    const int ntile = (c.f0_hi - c.f0_lo + 15) / 16;
    k_sync<<<dim3(ntile, B), 256, sync_lds_bytes(c), h->stream>>>(h->d_grid, h->d_best_score, h->d_best_h0, c);
    k_topk<<<B, 1024, 0, h->stream>>>(h->d_best_score, h->d_best_h0, h->d_rec, h->d_ncand, c);
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

int ft8rx_llr_grid(ft8rx_handle* h, const float* grid, int B, int n, const int32_t* frame, const int32_t* f0_idx,
                   const int32_t* h0_idx, float* llr, float* sd, int32_t* snr) {
    if (!h || !grid || B < 1 || B > h->max_frames || n < 1) return -1;
    HIPCHK(h, hipSetDevice(h->device));
    HIPCHK(h, hipMemcpy(h->d_grid, grid, sizeof(float) * (size_t)B * FT8RX_GRID_ROWS * FT8RX_GRID_COLS, hipMemcpyHostToDevice));
    std::vector<int32_t> trip(3 * (size_t)n);
    for (int i = 0; i < n; i++) { trip[3 * i] = frame[i]; trip[3 * i + 1] = f0_idx[i]; trip[3 * i + 2] = h0_idx[i]; }
    Scratch S{h};
    int32_t* d_trip = S.put(trip.data(), trip.size()); NEED(d_trip);
    float* d_llr = S.get<float>((size_t)n * 174); NEED(d_llr);
    float* d_sd = S.get<float>(n); NEED(d_sd);
    int32_t* d_snr = S.get<int32_t>(n); NEED(d_snr);
    k_grid_llr<<<n, 64, 0, h->stream>>>(h->d_grid, nullptr, nullptr, d_llr, h->cfg, d_trip, d_sd, d_snr);
    HIPCHK(h, hipStreamSynchronize(h->stream));
    HIPCHK(h, hipMemcpy(llr, d_llr, sizeof(float) * (size_t)n * 174, hipMemcpyDeviceToHost));
    HIPCHK(h, hipMemcpy(sd, d_sd, sizeof(float) * n, hipMemcpyDeviceToHost));
    HIPCHK(h, hipMemcpy(snr, d_snr, sizeof(int32_t) * n, hipMemcpyDeviceToHost));
    return 0;
}

int ft8rx_cycle_spectrum(ft8rx_handle* h, const int16_t* audio, int B, float* spec) {
    if (!h || !audio || !spec || B < 1 || B > h->max_frames) return -1;
    HIPCHK(h, hipSetDevice(h->device));
    HIPCHK(h, hipMemcpy(h->d_audio, audio, sizeof(int16_t) * (size_t)B * FT8RX_NSAMP, hipMemcpyHostToDevice));
    k_cyc_a<<<dim3(40, B), 256, 0, h->stream>>>(h->d_audio, h->d_A, h->T);
    k_cyc_b<<<dim3(75, B), 256, 0, h->stream>>>(h->d_A, h->d_Z, h->T);
    k_cyc_c<<<dim3(FT8RX_SPEC_BINS / 256, B), 256, 0, h->stream>>>(h->d_Z, h->d_spec, h->T);
    HIPCHK(h, hipStreamSynchronize(h->stream));
    HIPCHK(h, hipMemcpy(spec, h->d_spec, sizeof(cpx) * (size_t)B * FT8RX_SPEC_BINS, hipMemcpyDeviceToHost));
    return 0;
}

int ft8rx_fine(ft8rx_handle* h, const float* spec, int B, int n, const int32_t* frame, const int32_t* f0_idx, const int32_t* h0_idx,
               int32_t* ret, int32_t* ttweak, int32_t* ftweak, int32_t* nsync, float* llr, float* sd, int32_t* snr, float* sgrid) {
    if (!h || !spec || B < 1 || B > h->max_frames || n < 1) return -1;
    HIPCHK(h, hipSetDevice(h->device));
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
    k_fine<<<n, FINE_NT, 0, h->stream>>>(h->d_spec, nullptr, nullptr, d_llr, h->T, h->cfg, d_trip, d_out, d_sd, d_sg);
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
    HIPCHK(h, hipSetDevice(h->device));
    Scratch S{h};
    float* d_in = S.put(llr, (size_t)n * 174); NEED(d_in);
    float* d_out = S.get<float>((size_t)n * 174); NEED(d_out);
    HIPCHK(h, hipMemset(d_out, 0, sizeof(float) * (size_t)n * 174));
    Att* d_att = S.get<Att>(n); NEED(d_att);
    k_bp<<<n, 64, 0, h->stream>>>(2, d_in, nullptr, nullptr, nullptr, d_att, d_out, nullptr, nullptr, h->cfg, max_ncheck0, max_iters);
    HIPCHK(h, hipStreamSynchronize(h->stream));
    std::vector<Att> a(n);
    HIPCHK(h, hipMemcpy(a.data(), d_att, sizeof(Att) * n, hipMemcpyDeviceToHost));
    for (int i = 0; i < n; i++) { ok[i] = a[i].ok; msg_lo[i] = a[i].lo; msg_hi[i] = a[i].hi; n_its[i] = a[i].ok ? a[i].n_its : -1; has_out[i] = a[i].has_out; }
    if (llr_out) HIPCHK(h, hipMemcpy(llr_out, d_out, sizeof(float) * (size_t)n * 174, hipMemcpyDeviceToHost));
    return 0;
}

int ft8rx_osd(ft8rx_handle* h, const float* llr, int n, int singleflips, int doubleflips, int32_t* ok, uint64_t* msg_lo,
              uint64_t* msg_hi, int32_t* trial) {
    if (!h || !llr || n < 1) return -1;
    if (singleflips < 0 || singleflips > 64 || doubleflips < 0 || 1 + singleflips + singleflips * doubleflips > OSD_MAXTRIALS) { set_err(h, "ft8rx_osd: flip counts out of range"); return -1; }
    HIPCHK(h, hipSetDevice(h->device));
    Scratch S{h};
    float* d_in = S.put(llr, (size_t)n * 174); NEED(d_in);
    Att* d_att = S.get<Att>(n); NEED(d_att);
    k_osd<<<n, 64, 0, h->stream>>>(2, d_in, nullptr, nullptr, nullptr, nullptr, d_att, nullptr, nullptr, singleflips, doubleflips);
    HIPCHK(h, hipStreamSynchronize(h->stream));
    std::vector<Att> a(n);
    HIPCHK(h, hipMemcpy(a.data(), d_att, sizeof(Att) * n, hipMemcpyDeviceToHost));
    for (int i = 0; i < n; i++) { ok[i] = a[i].ok; msg_lo[i] = a[i].lo; msg_hi[i] = a[i].hi; trial[i] = a[i].ok ? a[i].n_its : -1; }
    return 0;
}

int ft8rx_crc_valid(ft8rx_handle* h, const float* cw91, int n, int32_t* res, uint64_t* msg_lo, uint64_t* msg_hi) {
    if (!h || !cw91 || n < 1) return -1;
    HIPCHK(h, hipSetDevice(h->device));
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
    HIPCHK(h, hipSetDevice(h->device));
    Scratch S{h};
    uint64_t* d_lo = S.put(msg_lo, n); NEED(d_lo);
    uint64_t* d_hi = S.put(msg_hi, n); NEED(d_hi);
    int32_t* d_v = S.get<int32_t>(n); NEED(d_v);
    k_valid_probe<<<(n + 255) / 256, 256, 0, h->stream>>>(d_lo, d_hi, n, d_v);
    HIPCHK(h, hipStreamSynchronize(h->stream));
    HIPCHK(h, hipMemcpy(valid, d_v, sizeof(int32_t) * n, hipMemcpyDeviceToHost));
    return 0;
}

int16_t* ft8rx_staging_audio(ft8rx_handle* h) { return h ? h->d_audio : nullptr; }

int ft8rx_copy_to_host(ft8rx_handle* h, void* dst, const void* d_src, uint64_t bytes) {
    if (!h || !dst || !d_src) return -1;
    HIPCHK(h, hipSetDevice(h->device));
    HIPCHK(h, hipMemcpy(dst, d_src, (size_t)bytes, hipMemcpyDeviceToHost));
    return 0;
}

int ft8rx_synth_frames(ft8rx_handle* h, uint64_t seed, int first_index, int n_frames, int n_signals,
                       const void* signal_table, int signal_bytes, const double* pulse_cumsum, int16_t* d_audio) {
    if (!h || !signal_table || !pulse_cumsum || !d_audio || n_frames < 1 || n_signals < 0 || n_signals > 64) return -1;
    if (signal_bytes != (int)sizeof(SynthSig)) { set_err(h, "ft8rx_synth_frames: signal record is %d bytes, expected %d", signal_bytes, (int)sizeof(SynthSig)); return -1; }
    HIPCHK(h, hipSetDevice(h->device));
    Scratch S{h};
    SynthSig* d_s = S.put((const SynthSig*)signal_table, (size_t)n_frames * (n_signals ? n_signals : 1)); NEED(d_s);
    double* d_q = S.put(pulse_cumsum, 5761); NEED(d_q);
    k_synth<<<dim3((FT8RX_NSAMP / 4 + 255) / 256, n_frames), 256, 0, h->stream>>>(d_audio, d_s, n_signals, d_q, (uint32_t)seed, (uint32_t)(seed >> 32), first_index);
    HIPCHK(h, hipStreamSynchronize(h->stream));
    return 0;
}

int ft8rx_package_batch(const ft8rx_record* records, const int32_t* counts, const ft8rx_event* events, const int32_t* event_counts,
                        int n_frames, int max_cands, ft8rx_message* out, int max_msgs, int32_t* out_counts, int n_threads) {
    if (!records || !counts || !events || !event_counts || !out || !out_counts || n_frames < 1 || max_cands < 1 || max_msgs < 1) return -1;
    if (n_threads < 1) n_threads = 1;
    if (n_threads > n_frames) n_threads = n_frames;
    auto work = [&](int t) {
        for (int f = t; f < n_frames; f += n_threads) {
            int nev = event_counts[f] < FT8RX_EVENT_CAP ? event_counts[f] : FT8RX_EVENT_CAP;
            out_counts[f] = hostmsg::package_frame(records + (size_t)f * max_cands, counts[f], events + (size_t)f * FT8RX_EVENT_CAP, nev,
                                                   out + (size_t)f * max_msgs, max_msgs);
        }
    };
    if (n_threads == 1) { work(0); return 0; }
    std::vector<std::thread> pool;
    for (int t = 0; t < n_threads; t++) pool.emplace_back(work, t);
    for (auto& th : pool) th.join();
    return 0;
}

int ft8rx_math_probe(ft8rx_handle* h, int which, const float* x, int n, float* y) {
    if (!h || !x || !y || n < 1) return -1;
    HIPCHK(h, hipSetDevice(h->device));
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
