// llr.hpp -- LLR extraction from the search grid, AP masks (receiver.py:208-222, 109-117)
// Part of libft8rx.so.  llr_from_p is shared with the second translation unit (k_fine); the rest is the main unit's.
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

#ifndef FT8RX_ILP_UNIT
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

// ------------------------------------------------------------------------------------ ipass-0 pre-check
// What every ipass-0 attempt starts with, for the five AP variants of one candidate, while its LLRs are still in LDS:
// GOOD91 (receiver.py:119-122: CRC + unpack on the hard decisions of llr[:91]) and BP's initial unsatisfied-check count
// (decoders.py:157-159).  Most attempts end right there (count > bp_nc0_a), and a wavefront of its own per attempt cost ~12 us of
// dependent loads for that -- 78 % of the first BP launch.  Resolved attempts get their final Att record here; the others are
// marked pending (pad[1] = 1) and k_worklist_att puts them on k_bp's attempt list.
struct ChkMasks { uint64_t m[6]; };          // this lane's two checks (lane, lane + 64) as membership masks over the 174 variables
FT8_DEV ChkMasks chk_masks(int lane) {
    ChkMasks c;
    c.m[0] = d_CHK_MASK[lane][0]; c.m[1] = d_CHK_MASK[lane][1]; c.m[2] = d_CHK_MASK[lane][2];
    c.m[3] = d_CHK_MASK[lane + 64][0]; c.m[4] = d_CHK_MASK[lane + 64][1]; c.m[5] = d_CHK_MASK[lane + 64][2];
    return c;
}
FT8_DEV void bp0_precheck(int lane, const float* llr /*LDS [174]*/, const ChkMasks& cm, int frame, int ci,
                          Att* __restrict__ att /*this candidate's [5]*/, ft8rx_event* ev, int32_t* evcount, int max_nc0, int max_iters) {
    const float v0 = llr[lane], v1 = llr[64 + lane], v2 = llr[128 + (lane < 46 ? lane : 0)];
    const uint64_t m27 = (1ull << 27) - 1;
    uint64_t h0[5], h1[5];
    int nchk[5];
#pragma unroll
    for (int ap = 0; ap < 5; ap++) {
        h0[ap] = __ballot(ap_value(ap, lane, v0) > 0.0f); h1[ap] = __ballot(ap_value(ap, 64 + lane, v1) > 0.0f);
        const uint64_t h2 = __ballot(lane < 46 && ap_value(ap, 128 + lane, v2) > 0.0f);
        const int par0 = (__popcll(h0[ap] & cm.m[0]) + __popcll(h1[ap] & cm.m[1]) + __popcll(h2 & cm.m[2])) & 1;
        const int par1 = (__popcll(h0[ap] & cm.m[3]) + __popcll(h1[ap] & cm.m[4]) + __popcll(h2 & cm.m[5])) & 1;
        nchk[ap] = __popcll(__ballot(par0)) + __popcll(__ballot(par1));
    }
    // CRC syndromes of the five words with two load instructions: row g of the wavefront (16 lanes) takes variant g, variant 4
    // rides in the upper half of row 0's registers
    const int g = lane >> 4;
    const uint64_t w0 = g == 0 ? h0[0] : g == 1 ? h0[1] : g == 2 ? h0[2] : h0[3];
    const uint64_t w1 = g == 0 ? h1[0] : g == 1 ? h1[1] : g == 2 ? h1[2] : h1[3];
    const unsigned t = ft8_xor_row16(ft8_crc_entry(w0, w1 & m27, lane) | (ft8_crc_entry(h0[4], h1[4] & m27, lane) << 16));
    // lanes 0..4 finish one variant each (the slow path -- CRC match: unpack + validity, event -- is ordinary per-lane code)
    const unsigned syn_lo = (unsigned)__builtin_amdgcn_ds_bpermute((lane & 3) << 6, (int)t) & 0xFFFFu;     // lane l < 4 <- row l
    const unsigned syn4 = (unsigned)__builtin_amdgcn_readfirstlane((int)t) >> 16;
    // (the ladder takes the variants in order and stops at the first success, receiver.py:72-78: a GOOD91 success at variant g
    // means the variants after g are never tried -- no event, not pending)
    int r = 0;
    uint64_t lo = 0, hi = 0;
    int myn = 0;
    if (lane < 5) {
        const unsigned syn = lane < 4 ? syn_lo : syn4;
        const uint64_t my0 = lane == 0 ? h0[0] : lane == 1 ? h0[1] : lane == 2 ? h0[2] : lane == 3 ? h0[3] : h0[4];
        const uint64_t my1 = lane == 0 ? h1[0] : lane == 1 ? h1[1] : lane == 2 ? h1[2] : lane == 3 ? h1[3] : h1[4];
        myn = lane == 0 ? nchk[0] : lane == 1 ? nchk[1] : lane == 2 ? nchk[2] : lane == 3 ? nchk[3] : nchk[4];
        if (syn == 0) r = ft8_crc_check(my0, my1 & m27, &lo, &hi);
    }
    const uint64_t okm = __ballot(r == 2);
    const int first = okm ? __builtin_ctzll(okm) : 5;
    if (lane < 5) {
        Att a; memset(&a, 0, sizeof(a)); a.n_its = -1;
        if (lane <= first) {
            if (r) log_event(ev, evcount, frame, ci, 0, lane, 0, lo, hi, r == 2);
            if (r == 2) { a.ok = 1; a.lo = lo; a.hi = hi; a.n_its = 0; a.method = FT8RX_M_GOOD91; }
            else if (max_iters > 0 && myn > max_nc0) a.nc0 = (uint8_t)myn;        // BP gives up before its first iteration
            else a.pad[1] = 1;                                                    // pending: k_bp runs it
        }
        att[lane] = a;
    }
}

// block of 64 = one candidate (or one test triple when `trip` is given)
__global__ __launch_bounds__(64) void k_grid_llr(const float* __restrict__ grid, ft8rx_record* __restrict__ rec,
                                                 const int32_t* __restrict__ ncand, float* __restrict__ llr0,
                                                 ft8rx_config cfg, const int32_t* __restrict__ trip, float* __restrict__ t_sd,
                                                 int32_t* __restrict__ t_snr, Att* __restrict__ att0, ft8rx_event* ev,
                                                 int32_t* evcount, int B) {
    __shared__ float p[464];
    __shared__ float llr[174];
    __shared__ float sq[174];
    const int lane = threadIdx.x;
    int frame, ci, f0, h0;
    size_t slot = blockIdx.x;                   // where this block's outputs go: the triple's index, or the candidate's
    if (trip) { frame = trip[3 * blockIdx.x]; f0 = trip[3 * blockIdx.x + 1]; h0 = trip[3 * blockIdx.x + 2]; ci = 0; }
    else {
        if (!xcd_frame_map(blockIdx.x, cfg.max_cands, B, frame, ci)) return;      // a frame's candidates gather from one XCD's L2 (launch: XCD_GRID(B, max_cands))
        slot = (size_t)frame * MAXC + ci;
        if (ci >= ncand[frame]) return;
        const ft8rx_record& r = rec[(size_t)frame * MAXC + ci];
        f0 = r.f0_idx; h0 = r.h0_idx;
    }
    ChkMasks cm;
    if (att0) cm = chk_masks(lane);              // issued with the gather below
    const float* g = grid + (size_t)frame * FT8RX_GRID_ROWS * FT8RX_GRID_COLS;
    {
        float v[8];
#pragma unroll
        for (int q = 0; q < 8; q++) {                  // all eight gathers of a lane in flight together (a load->store loop: 129 -> 105 us)
            const int i = lane + 64 * q, ic = i < 464 ? i : 0;
            v[q] = grid_at(g, h0 + 4 + 4 * (int)d_PAYSYM[ic >> 3], f0 + 1 + 2 * (ic & 7));        // receiver.py:358-362
        }
#pragma unroll
        for (int q = 0; q < 8; q++) if (lane + 64 * q < 464) p[lane + 64 * q] = v[q];
    }
    __syncthreads();
    float sd; int snr;
    llr_from_p(p, llr, sq, lane, true, &sd, &snr);
    float* out = llr0 + slot * 174;
    for (int i = lane; i < 174; i += 64) out[i] = llr[i];
    if (lane == 0) {
        if (trip) { t_sd[blockIdx.x] = sd; t_snr[blockIdx.x] = snr; }
        else {
            ft8rx_record& r = rec[(size_t)frame * MAXC + ci];
            r.grid_sd = sd; r.snr_grid = (int8_t)snr;
            if (sd <= cfg.llr_sd_min) r.status = FT8RX_ST_STOP_GRID_SD;
        }
    }
    if (att0 && !(sd <= cfg.llr_sd_min))        // pipeline: the candidate stays ACTIVE -> pre-check its five ipass-0 attempts
        bp0_precheck(lane, llr, cm, frame, ci, att0 + slot * 5, ev, evcount, cfg.bp_nc0_a, cfg.bp_iters_a);
}
#endif  // FT8RX_ILP_UNIT

#endif
