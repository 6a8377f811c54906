// sync_search.hpp -- K2/K3: Costas sync search and top-K candidate selection (receiver.py:338-367)
// Part of libft8rx.so; included by ft8rx.hip (single translation unit: the kernels share __constant__/__device__ tables).
#ifndef FT8RX_SYNC_SEARCH_HPP
#define FT8RX_SYNC_SEARCH_HPP

// ------------------------------------------------------------------------------------ K2 sync search
// block = 16 consecutive f0 of one frame; LDS tile = every grid row any h0 can touch x 29 columns.
// accumulate != 0: a later window of a wide search_time_range (ft8rx.hip: launch_sync) -- the stored result of the earlier windows stays
// unless this window holds a strictly larger score (windows come in ascending h0: the first strict maximum of the whole range).
template <bool accumulate>
FT8_DEV void sync_block(const float* __restrict__ grid, float* __restrict__ best_score, int32_t* __restrict__ best_h0, const ft8rx_config& cfg) {
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
    // tile load: rows outside 1..375 read the grid's initial 1.0 (receiver.py:240), columns beyond the grid 0 -- resolved by integer
    // masks on an always-valid (clamped) load, so the loop body is straight-line and the loads of successive iterations overlap
    for (int i = tid; i < nrows * 29; i += 256) {
        const int r = i / 29, c = i - r * 29;
        const int col = f0base + c;
        const bool incol = col < FT8RX_GRID_COLS;
        const uint32_t raw = __float_as_uint(grid_at(g, rlo + r, incol ? col : 0));
        tile[i] = __uint_as_float(raw & (incol ? 0xFFFFFFFFu : 0u));
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
            const size_t o = (size_t)f * NF0MAX + (f0 - cfg.f0_lo);
            if (!accumulate || bs > best_score[o]) { best_score[o] = bs; best_h0[o] = h; }
        }
    }
}

__global__ __launch_bounds__(256) void k_sync(const float* __restrict__ grid, float* __restrict__ best_score,
                                              int32_t* __restrict__ best_h0, ft8rx_config cfg) { sync_block<false>(grid, best_score, best_h0, cfg); }
// a later window of a search_time_range wider than SYNC_WIN offsets
__global__ __launch_bounds__(256) void k_sync_acc(const float* __restrict__ grid, float* __restrict__ best_score,
                                                  int32_t* __restrict__ best_h0, ft8rx_config cfg) { sync_block<true>(grid, best_score, best_h0, cfg); }

// ------------------------------------------------------------------------------------ K3 top-K
// threshold, stable sort by score descending (ties: f0 ascending = original order), keep max_cands.
// NF0MAX keys (1024; 2048 in the wide build) on 1024 threads: bitonic network in LDS, KPT compare-exchanges per thread and step.
__global__ __launch_bounds__(1024) void k_topk(const float* __restrict__ best_score, const int32_t* __restrict__ best_h0,
                                               ft8rx_record* __restrict__ rec, int32_t* __restrict__ ncand, ft8rx_config cfg,
                                               int32_t* __restrict__ evcount, int32_t* __restrict__ wcount,
                                               const uint8_t* __restrict__ colmask) {
    constexpr int KPT = NF0MAX / 1024;
    // the chain's per-frame event counters and its work-list lengths start at zero: cleared here (the first kernel after which they
    // are used) instead of by two memset launches per chain
    if (evcount && threadIdx.x == 0) evcount[blockIdx.x] = 0;
    if (wcount && blockIdx.x == 0 && threadIdx.x < WL_N) wcount[threadIdx.x] = 0;
    __shared__ uint64_t key[NF0MAX];
    const int f = blockIdx.x, tid = threadIdx.x;
    const int nf0 = cfg.f0_hi - cfg.f0_lo;
#pragma unroll
    for (int q = 0; q < KPT; q++) {
        const int i = tid + 1024 * q;
        uint64_t k = ~0ull;
        if (i < nf0) {
            float s = best_score[(size_t)f * NF0MAX + i];
            // local re-search (ft8rx_set_search_mask; receiver_sub.py:434-445): only the masked columns, every score above 0
            const bool take = colmask ? (colmask[(size_t)f * NF0MAX + i] != 0 && s > 0.0f) : (s > cfg.sync_score_min);
            if (take) k = ((uint64_t)(~__float_as_uint(s)) << 32) | (uint32_t)i;   // s > 0: bit pattern is monotonic
        }
        key[i] = k;
    }
    __syncthreads();
    for (int size = 2; size <= NF0MAX; size <<= 1) {
        for (int stride = size >> 1; stride > 0; stride >>= 1) {
#pragma unroll
            for (int q = 0; q < KPT; q++) {
                const int i = tid + 1024 * q, partner = i ^ stride;
                if (partner > i) {
                    uint64_t a = key[i], b = key[partner];
                    bool up = ((i & size) == 0);
                    if ((a > b) == up) { key[i] = b; key[partner] = a; }
                }
            }
            __syncthreads();
        }
    }
    int mine = 0;
#pragma unroll
    for (int q = 0; q < KPT; q++) mine += key[tid + 1024 * q] != ~0ull;
    __shared__ int tot;
    if (tid == 0) tot = 0;
    __syncthreads();
    if (mine) atomicAdd(&tot, mine);
    __syncthreads();
    const int cnt = tot;
    if (tid == 0) ncand[f] = cnt < cfg.max_cands ? cnt : cfg.max_cands;
#pragma unroll
    for (int q = 0; q < KPT; q++) {                                   // rank = position in the sorted keys (max_cands <= MAXC <= NF0MAX)
        const int rank = tid + 1024 * q;
        if (rank >= cfg.max_cands) break;
        const uint64_t kk = key[rank];
        ft8rx_record r; memset(&r, 0, sizeof(r));
        if (kk != ~0ull) {
            int i = (int)(kk & 0xffffffffu);
            r.f0_idx = (int16_t)(cfg.f0_lo + i);
            r.h0_idx = (int16_t)best_h0[(size_t)f * NF0MAX + i];
            r.score = __uint_as_float(~(uint32_t)(kk >> 32));
            r.status = FT8RX_ST_ACTIVE; r.ipass = 0xff;
        } else r.status = FT8RX_ST_EXHAUSTED;
        rec[(size_t)f * MAXC + rank] = r;
    }
}

#endif
