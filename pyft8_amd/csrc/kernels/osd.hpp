// osd.hpp -- ordered-statistics decoding (decoders.py:223-272)
// Part of libft8rx.so; included by ft8rx.hip (single translation unit: the kernels share __constant__/__device__ tables).
#ifndef FT8RX_OSD_HPP
#define FT8RX_OSD_HPP

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

#endif
