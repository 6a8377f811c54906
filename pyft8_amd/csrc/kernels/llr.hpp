// llr.hpp -- LLR extraction from the search grid, AP masks (receiver.py:208-222, 109-117)
// Part of libft8rx.so; included by ft8rx.hip (single translation unit: the kernels share __constant__/__device__ tables).
#ifndef FT8RX_LLR_HPP
#define FT8RX_LLR_HPP

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

#endif
