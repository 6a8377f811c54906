// osd.hpp -- ordered-statistics decoding (decoders.py:223-272)
// Part of libft8rx.so; included by ft8rx.hip (single translation unit: the kernels share __constant__/__device__ tables).
#ifndef FT8RX_OSD_HPP
#define FT8RX_OSD_HPP

// ------------------------------------------------------------------------------------ OSD (decoders.py:223-272)
// One wavefront per attempt, three phases:
//   1. reliability order: LDS bitonic network over 256 composite keys (fixed tie rule for the reference's unstable argsort);
//   2. most-reliable-basis Gauss-Jordan over GF(2) with the generator held COLUMN-wise: lane l owns columns l, 64+l, 128+l
//      of G0 (91 row bits each, 3 x u32).  A visited column is broadcast to scalar registers; "independent of the accepted
//      columns" is then a scalar test (any 1 in an unlocked row), the pivot row a scalar find-first-set, and the elimination
//      one masked XOR per owned column -- no ballots, no cross-lane shuffles, nothing on the dependent chain but readlanes;
//   3. trials: CRC-14 is linear, so each trial's syndrome is the XOR of the precomputed syndromes of the order-0 codeword
//      and of its flip rows; a lane tests one trial with three 16-bit LDS reads.  Only zero-syndrome trials (2^-14 of them)
//      rebuild the codeword, run the validity predicate and log the reference's unpack() call.
// The trial list (order 0, single flips, the reference's restricted double flips, then the build's order-3 extension) is a
// table built by the host from the configuration, in the reference's trial order (decoders.py:248-272).
FT8_DEV uint64_t shfl64(uint64_t v, int src) {
    uint32_t lo = __shfl((uint32_t)v, src), hi = __shfl((uint32_t)(v >> 32), src);
    return ((uint64_t)hi << 32) | lo;
}

#define OSD_MAXFLIP 64            /* flip rows kept per attempt (singles / order-3 depth <= 64) */
#define OSD_MAXTRIALS 16384       /* trial index must fit the 16-bit seq of the event log */
#define OSD_NONE 0xFFu            /* "no flip" in a packed trial entry (i | j << 8 | k << 16) */

__device__ uint32_t d_G0T[192][3];       // column v of G0 = [I | A^T]: row bits 0..31, 32..63, 64..90 (columns >= 174 are zero)
FT8_DEV unsigned osd_syndrome(uint64_t w0, uint64_t w1) { return ft8_crc_syndrome(w0, w1); }     // table d_CRC_T: ft8_dev.h

// mode 0: pipeline (work = (candidate, slot 0..9)); mode 2: raw vectors
__global__ __launch_bounds__(64) void k_osd(int mode, const float* __restrict__ llr_in, const float* __restrict__ saved,
                                            const Att* __restrict__ attB, ft8rx_record* __restrict__ rec,
                                            const int32_t* __restrict__ ncand, Att* __restrict__ attO,
                                            ft8rx_event* ev, int32_t* evcount, const uint32_t* __restrict__ trials, int ntr,
                                            int nflip, int max_hd) {
    __shared__ float llr[176];
    __shared__ uint64_t skey[256];
    __shared__ uint64_t flip[OSD_MAXFLIP + 1][3];          // [OSD_MAXFLIP] = 0: the "no flip" row
    __shared__ uint16_t fsyn[OSD_MAXFLIP + 2];
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
#ifndef OSD_TIMING_SKIP_SORT            /* timing-only builds (tools/ab_variants.sh): never defined in the product */
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
#endif
    const int ord0 = (int)(uint32_t)skey[lane], ord1 = (int)(uint32_t)skey[64 + lane], ord2 = (lane < 46) ? (int)(uint32_t)skey[128 + lane] : 0;
    const uint64_t hard0 = __ballot(llr[lane] > 0.0f), hard1 = __ballot(llr[64 + lane] > 0.0f),
                   hard2 = __ballot(lane < 46 && llr[128 + (lane < 46 ? lane : 0)] > 0.0f);
    // ---- Gauss-Jordan, generator held column-wise.  Column c of the current matrix is a 91-bit vector over the rows; a row is
    // "locked" once it has been made the unit row of an accepted column.  Visiting column c (in reliability order, exactly as
    // decoders.py:228-242): it is independent of the accepted columns iff it has a 1 in an unlocked row; the lowest such row r
    // becomes its pivot (which row is picked does not change the result: the reduced matrix for a given basis is unique up to
    // the row labels, and the codeword / flip rows below are label-free); clearing the other 1s of column c = adding row r to
    // those rows = XORing (column c minus bit r) into every column that has a 1 in row r.  A column that already is a unit
    // column needs no update at all (the systematic columns that are still untouched: about 40 % of the basis).
    // lock / hm (locked rows whose accepted column has hard decision 1) / the visited column live in scalar registers.
    uint32_t x00 = d_G0T[lane][0], x01 = d_G0T[lane][1], x02 = d_G0T[lane][2];
    uint32_t x10 = d_G0T[64 + lane][0], x11 = d_G0T[64 + lane][1], x12 = d_G0T[64 + lane][2];
    uint32_t x20 = d_G0T[128 + lane][0], x21 = d_G0T[128 + lane][1], x22 = d_G0T[128 + lane][2];
    uint32_t lock0 = 0, lock1 = 0, lock2 = ~((1u << 27) - 1u), hm0 = 0, hm1 = 0, hm2 = 0;
    int prowA = 0, prowB = 0;                  // prow[k] = row locked by the k-th accepted column: lane k of prowA (k < 64) / lane k - 64 of prowB
    int k = 0;
#ifdef OSD_TIMING_SKIP_ELIM
    k = 91;
#endif
    for (int ic = 0; ic < 174 && k < 91; ic++) {
        const int col = __builtin_amdgcn_readlane((ic < 64) ? ord0 : ((ic < 128) ? ord1 : ord2), ic & 63);
        const int sel = col >> 6, cl = col & 63;
        uint32_t c0, c1, c2;
        if (sel == 0) { c0 = __builtin_amdgcn_readlane(x00, cl); c1 = __builtin_amdgcn_readlane(x01, cl); c2 = __builtin_amdgcn_readlane(x02, cl); }
        else if (sel == 1) { c0 = __builtin_amdgcn_readlane(x10, cl); c1 = __builtin_amdgcn_readlane(x11, cl); c2 = __builtin_amdgcn_readlane(x12, cl); }
        else { c0 = __builtin_amdgcn_readlane(x20, cl); c1 = __builtin_amdgcn_readlane(x21, cl); c2 = __builtin_amdgcn_readlane(x22, cl); }
        const uint32_t a0 = c0 & ~lock0, a1 = c1 & ~lock1, a2 = c2 & ~lock2;
        if (!(a0 | a1 | a2)) continue;                        // dependent on the accepted columns
        const int r = a0 ? __builtin_ctz(a0) : (a1 ? 32 + __builtin_ctz(a1) : 64 + __builtin_ctz(a2));
        const uint32_t bit = 1u << (r & 31);
        const uint32_t b0 = (r < 32) ? bit : 0u, b1 = (r >= 32 && r < 64) ? bit : 0u, b2 = (r >= 64) ? bit : 0u;
        const uint32_t m0 = c0 & ~b0, m1 = c1 & ~b1, m2 = c2 & ~b2;
        if (m0 | m1 | m2) {
            if ((x00 & b0) | (x01 & b1) | (x02 & b2)) { x00 ^= m0; x01 ^= m1; x02 ^= m2; }
            if ((x10 & b0) | (x11 & b1) | (x12 & b2)) { x10 ^= m0; x11 ^= m1; x12 ^= m2; }
            if ((x20 & b0) | (x21 & b1) | (x22 & b2)) { x20 ^= m0; x21 ^= m1; x22 ^= m2; }
        }
        lock0 |= b0; lock1 |= b1; lock2 |= b2;
        const uint64_t hw = (sel == 0) ? hard0 : ((sel == 1) ? hard1 : hard2);
        if ((hw >> cl) & 1ull) { hm0 |= b0; hm1 |= b1; hm2 |= b2; }
        if (lane == (k & 63)) { if (k < 64) prowA = r; else prowB = r; }
        k++;
    }
    // order-0 codeword: bit v = parity of (column v AND hm) -- the XOR of the locked rows whose accepted column has hard bit 1
    const uint64_t cw0 = __ballot((__popc(x00 & hm0) + __popc(x01 & hm1) + __popc(x02 & hm2)) & 1),
                   cw1 = __ballot((__popc(x10 & hm0) + __popc(x11 & hm1) + __popc(x12 & hm2)) & 1),
                   cw2 = __ballot((__popc(x20 & hm0) + __popc(x21 & hm1) + __popc(x22 & hm2)) & 1);
    // flip rows: flip[i] = the row locked by accepted column 90 - i (the least reliable basis members first), as 174 column bits
#ifdef OSD_TIMING_SKIP_FLIPS
    nflip = 0;
#endif
    for (int i = 0; i < nflip; i++) {
        const int kk = 90 - i;
        uint64_t f0 = 0, f1 = 0, f2 = 0;
        if (kk >= 0 && kk < k) {
            const int r = (kk < 64) ? __builtin_amdgcn_readlane(prowA, kk) : __builtin_amdgcn_readlane(prowB, kk - 64);
            const uint32_t bit = 1u << (r & 31);
            if (r < 32)      { f0 = __ballot(x00 & bit); f1 = __ballot(x10 & bit); f2 = __ballot(x20 & bit); }
            else if (r < 64) { f0 = __ballot(x01 & bit); f1 = __ballot(x11 & bit); f2 = __ballot(x21 & bit); }
            else             { f0 = __ballot(x02 & bit); f1 = __ballot(x12 & bit); f2 = __ballot(x22 & bit); }
        }
        if (lane == 0) { flip[i][0] = f0; flip[i][1] = f1; flip[i][2] = f2; }
    }
    if (lane == 0) { flip[OSD_MAXFLIP][0] = 0; flip[OSD_MAXFLIP][1] = 0; flip[OSD_MAXFLIP][2] = 0; fsyn[OSD_MAXFLIP] = 0; }
    __syncthreads();
    const uint64_t M1 = (1ull << 27) - 1, M2 = (1ull << 46) - 1;
    if (lane < nflip) fsyn[lane] = (uint16_t)osd_syndrome(flip[lane][0], flip[lane][1] & M1);
    const unsigned syn_c = osd_syndrome(cw0, cw1 & M1);
    __syncthreads();
    Att res; memset(&res, 0, sizeof(res)); res.n_its = -1;
    const int ipass = (slot < 5) ? 5 : 6;
#ifdef OSD_TIMING_SKIP_TRIALS
    ntr = 0;
#endif
    for (int base = 0; base < ntr; base += 64) {
        const int t = base + lane;
        bool hit = false; int i = OSD_MAXFLIP, j = OSD_MAXFLIP, q = OSD_MAXFLIP;
        if (t < ntr) {
            const uint32_t e = trials[t];
            i = e & 0xFF; j = (e >> 8) & 0xFF; q = (e >> 16) & 0xFF;
            if (i == OSD_NONE) i = OSD_MAXFLIP;
            if (j == OSD_NONE) j = OSD_MAXFLIP;
            if (q == OSD_NONE) q = OSD_MAXFLIP;
            hit = (syn_c ^ fsyn[i] ^ fsyn[j] ^ fsyn[q]) == 0;
        }
        if (!__ballot(hit)) continue;                         // no CRC-consistent word among these 64 trials (the usual case)
        int r = 0, hd = 0; uint64_t lo = 0, hi = 0;
        if (hit) {
            const uint64_t w0 = cw0 ^ flip[i][0] ^ flip[j][0] ^ flip[q][0], w1 = cw1 ^ flip[i][1] ^ flip[j][1] ^ flip[q][1],
                           w2 = cw2 ^ flip[i][2] ^ flip[j][2] ^ flip[q][2];
            hd = __popcll(w0 ^ hard0) + __popcll(w1 ^ hard1) + __popcll((w2 ^ hard2) & M2);
            if (!(max_hd > 0 && hd > max_hd)) r = ft8_crc_check(w0, w1 & M1, &lo, &hi);     // gate (extension): no unpack() call beyond max_hd
        }
        const uint64_t acc = __ballot(r == 2);
        const int win = acc ? __builtin_ctzll(acc) : 64;
        if (r && lane <= win) log_event(ev, evcount, frame, ci, ipass, slot, t, lo, hi, r == 2);   // calls the reference made
        if (acc) {
            res.ok = 1; res.lo = shfl64(lo, win); res.hi = shfl64(hi, win); res.n_its = (int16_t)(base + win);
            res.method = (slot < 5) ? FT8RX_M_OSD : FT8RX_M_LDPC_B_OSD;
            res.pad[0] = (uint8_t)__shfl(hd, win);            // Hamming distance of the accepted codeword to the hard decisions
            break;
        }
    }
    if (lane == 0) attO[vec] = res;
}

#endif
