// osd.hpp -- ordered-statistics decoding (decoders.py:223-272)
// Part of libft8rx.so; included by ft8rx.hip (single translation unit: the kernels share __constant__/__device__ tables).
#ifndef FT8RX_OSD_HPP
#define FT8RX_OSD_HPP

// ------------------------------------------------------------------------------------ OSD (decoders.py:223-272)
// One wavefront per attempt, three phases:
//   1. reliability order: bitonic network over 256 composite keys held in registers, exchanged by DPP / ds_swizzle / ds_bpermute (fixed tie
//      rule for the reference's unstable argsort);
//   2. most-reliable-basis Gauss-Jordan over GF(2) with the generator held COLUMN-wise: lane l owns columns l, 64+l, 128+l
//      of G0 (91 row bits each, 3 x u32).  A visited column is broadcast to scalar registers; "independent of the accepted
//      columns" is then a scalar test (any 1 in an unlocked row), the pivot row a scalar find-first-set, and the elimination
//      one masked XOR per owned column -- no ballots, no cross-lane shuffles, nothing on the dependent chain but readlanes;
//   3. trials: CRC-14 is linear, so each trial's syndrome is the XOR of the precomputed syndromes of the order-0 codeword
//      and of its flip rows; a lane tests one trial with three 16-bit LDS reads.  Only zero-syndrome trials (2^-14 of them)
//      rebuild the codeword, run the validity predicate and log the reference's unpack() call.
// The trial list (order 0, single flips, the reference's restricted double flips, then the build's order-3 extension) is a
// table built by the host from the configuration, in the reference's trial order (decoders.py:248-272).
// Timing-only builds (-DOSD_TIMING, tools/osd_timing.py): lane 0 of every attempt accumulates the shader cycles between consecutive marks
// and adds them to g_osd_t[] at the end.  Never defined in the product.
#ifdef OSD_TIMING
__device__ unsigned long long g_osd_t[32768][10];       // per block: plain adds by lane 0 of its one wave, summed by the host (no atomics in the timed code)
#define OT_DECL unsigned long long ot_prev = __builtin_readcyclecounter();
#define OT(i) do { const unsigned long long ot_now = __builtin_readcyclecounter(); if (lane == 0) g_osd_t[blockIdx.x & 32767][i] += ot_now - ot_prev; ot_prev = __builtin_readcyclecounter(); } while (0)
#define OT_FLUSH do { if (lane == 0) g_osd_t[blockIdx.x & 32767][9] += 1ull; } while (0)
#else
#define OT_DECL
#define OT(i) do { } while (0)
#define OT_FLUSH do { } while (0)
#endif
FT8_DEV uint64_t shfl64(uint64_t v, int src) {
    uint32_t lo = __shfl((uint32_t)v, src), hi = __shfl((uint32_t)(v >> 32), src);
    return ((uint64_t)hi << 32) | lo;
}

// the value of lane ^ S: quad permutes (DPP) for S = 1, 2, ds_swizzle (crossbar only, no address register) for 4, 8, 16, ds_bpermute for 32
template <int S> FT8_DEV uint32_t osd_xlane(uint32_t v, int lane) {
    if (S == 1) return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1, 0xf, 0xf, false);          // quad_perm:[1,0,3,2]
    if (S == 2) return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x4E, 0xf, 0xf, false);          // quad_perm:[2,3,0,1]
    if (S == 4 || S == 8 || S == 16) return (uint32_t)__builtin_amdgcn_ds_swizzle((int)v, 0x1f | (S << 10));   // bit mode: lane ^ S within 32
    return __shfl(v, lane ^ S);
}
// one compare-exchange step of the bitonic network between lanes S apart: this lane keeps the smaller key iff keep_min
template <int S> FT8_DEV uint64_t osd_cx(uint64_t k, int lane, bool keep_min) {
    const uint64_t o = ((uint64_t)osd_xlane<S>((uint32_t)(k >> 32), lane) << 32) | osd_xlane<S>((uint32_t)k, lane);
    return ((k < o) == keep_min) ? k : o;
}
// the lane strides S, S/2, ..., 1 of one merge on NQ registers; up(q) = direction of block q
template <int S, int NQ, typename UP> FT8_DEV void osd_merge_lanes(uint64_t* kq, int lane, UP up) {
    const bool lower = (lane & S) == 0;
#pragma unroll
    for (int q = 0; q < NQ; q++) kq[q] = osd_cx<S>(kq[q], lane, lower == up(q));
    if constexpr (S > 1) osd_merge_lanes<S / 2, NQ>(kq, lane, up);
}

#define OSD_MAXFLIP 91            /* flip rows kept per attempt = all 91 basis positions (decoders.py:244-246 takes any count up to the basis size) */
#define OSD_FLIPS_A 62            /* flips 0..61: one bit each in the column's 64-bit word (bit 63 = order-0 codeword bit); flips 62..90: a second, 32-bit word
                                     that only exists for attempts with more than 62 flip rows (the reference's own callers use 30 / 40) */
#define OSD_MAXTRIALS 16384       /* trial index must fit the 16-bit seq of the event log */
#define OSD_NONE 0xFFu            /* "no flip" in a packed trial entry (i | j << 8 | k << 16) */

// CRC syndrome of codeword bit v alone (v < 77: message bit, 77..90: the CRC field bit itself), bit-sliced: d_SYNM[k][w] bit b = bit k
// of the syndrome of codeword bit 32 w + b.  Constant address space: wave-uniform reads become scalar loads (a uniform lookup in a
// 16-bit __device__ table is a broadcast vector load -- 91 of them per attempt kept the texture-address path busy, profiles/r02_notes.md)
__device__ __constant__ uint32_t d_SYNM[14][3];
__device__ uint32_t d_G0T[192][3];       // column v of G0 = [I | A^T]: row bits 0..31, 32..63, 64..90 (columns >= 174 are zero)
FT8_DEV unsigned osd_syndrome(uint64_t w0, uint64_t w1) { return ft8_crc_syndrome(w0, w1); }     // table d_CRC_T: ft8_dev.h

// mode 0: pipeline (work = (candidate, slot 0..9)); mode 2: raw vectors.  WIDE: more than 62 flip rows (k_osd_wide) -- a kernel of its
// own, because the second flip word costs 13 VGPRs = two of the seven waves per SIMD that hide this kernel's scalar-pipe latency
// (one kernel with a run-time switch: 0.737 -> 0.791 ms per 256 frames at the reference's 30 / 2)
template <bool WIDE>
FT8_DEV void osd_attempt(int lane, int mode, int bid, const float* __restrict__ llr_in, const float* __restrict__ saved,
                         const Att* __restrict__ attB, ft8rx_record* __restrict__ rec,
                         const int32_t* __restrict__ ncand, Att* __restrict__ attO,
                         ft8rx_event* ev, int32_t* evcount, const uint32_t* __restrict__ trials, int ntr,
                         int nflip, int max_hd) {
    __shared__ float llr[176];
    __shared__ uint64_t skey[256];
    __shared__ uint64_t ftab[192];                         // per column (natural order): bit i = flip i covers it (i < 62), bit 63 = order-0 codeword bit
    // the sort keys are dead once the reliability order has been read into registers; their 2 KB then hold
    uint32_t* ftabB = reinterpret_cast<uint32_t*>(skey);  // [192] bit i - 62 = flip i covers it (62 <= i < 91), and
    uint32_t* frow = ftabB + 192;                          // [3 (OSD_MAXFLIP + 1)] unit vectors of the flip columns (their pivot rows)
    static_assert((192 + 3 * (OSD_MAXFLIP + 1)) * sizeof(uint32_t) <= 256 * sizeof(uint64_t), "ftabB + frow overlay the sort keys");
    constexpr bool wide = WIDE;                            // nflip > OSD_FLIPS_A (the launcher picks the kernel)
    const int nflipA = wide ? OSD_FLIPS_A : nflip;
    __shared__ uint32_t hmw[3];
    __shared__ uint16_t fsyn[OSD_MAXFLIP + 2];             // [i] flip i, [OSD_MAXFLIP] = 0 ("no flip"), [OSD_MAXFLIP + 1] order-0 codeword
    int frame = 0, ci = 0, slot = 0; size_t vec = bid;
    OT_DECL
    if (mode == 0) {
        slot = bid % 10; int c = bid / 10; frame = c / MAXC; ci = c % MAXC;
        if (ci >= ncand[frame]) return;
        if (rec[(size_t)frame * MAXC + ci].status != FT8RX_ST_ACTIVE) return;
        if (slot >= 5 && !attB[(size_t)c * 5 + (slot - 5)].has_out) { if (lane == 0) { Att a; memset(&a, 0, sizeof(a)); a.n_its = -1; attO[(size_t)c * 10 + slot] = a; } return; }
        // slots 0..4: the fine LLRs with the AP override, slots 5..9: the saved BP outputs; three loads in flight either way
        const float* src = slot < 5 ? llr_in + (size_t)c * 174 : saved + ((size_t)c * 5 + (slot - 5)) * 174;
        const int apx = slot < 5 ? slot : 0;                                   // ap_value(0, ...) is the identity
        const float v0 = src[lane], v1 = src[64 + lane], v2 = src[128 + (lane < 46 ? lane : 0)];
        llr[lane] = ap_value(apx, lane, v0); llr[64 + lane] = ap_value(apx, 64 + lane, v1);
        if (lane < 46) llr[128 + lane] = ap_value(apx, 128 + lane, v2);
        vec = (size_t)c * 10 + slot;
    } else {
        for (int i = lane; i < 174; i += 64) llr[i] = llr_in[vec * 174 + i];
    }
    __syncthreads();
    // ---- reliability order: |llr| descending, ties and NaNs (last) by index (fixed rule for np.argsort, decoders.py:226).
    // Bitonic network over 256 composite keys ((~magnitude bits) << 32 | index), element 64 q + lane in register q of the lane: the steps
    // between lanes are cross-lane exchanges, the strides 64 / 128 are register pairs of one lane -- no LDS image, no barriers (the LDS form
    // spent 21 % of the kernel here, 36 barriers per attempt: profiles/r04_osd_timing.txt).  Register 3 would hold the 64 padding keys
    // (~0): every step that involves it is resolved by hand below.
    uint64_t kq[3];
#pragma unroll
    for (int q = 0; q < 3; q++) {
        const int i = lane + 64 * q;
        uint64_t key = ~0ull;
        if (i < 174) {
            const float x = llr[i];
            const uint32_t k32 = (x != x) ? 0u : ((__float_as_uint(x) & 0x7fffffffu) + 1u);
            key = ((uint64_t)(0xFFFFFFFFu - k32) << 32) | (uint32_t)i;
        }
        kq[q] = key;
    }
    OT(0);
#ifndef OSD_TIMING_SKIP_SORT            /* timing-only builds (tools/ab_variants.sh): never defined in the product */
    // sizes 2 .. 64: the three real blocks, each in its own register; block q ends ascending for even q, descending for odd q
    osd_merge_lanes<1, 3>(kq, lane, [&](int) { return (lane & 2) == 0; });
    osd_merge_lanes<2, 3>(kq, lane, [&](int) { return (lane & 4) == 0; });
    osd_merge_lanes<4, 3>(kq, lane, [&](int) { return (lane & 8) == 0; });
    osd_merge_lanes<8, 3>(kq, lane, [&](int) { return (lane & 16) == 0; });
    osd_merge_lanes<16, 3>(kq, lane, [&](int) { return (lane & 32) == 0; });
    osd_merge_lanes<32, 3>(kq, lane, [&](int q) { return (q & 1) == 0; });
    // size 128.  Blocks 0 (ascending) and 1 (descending) merge upwards: stride 64 is the register pair, then the lane strides.
    { const uint64_t a = kq[0], b = kq[1]; const bool lt = a < b; kq[0] = lt ? a : b; kq[1] = lt ? b : a; }
    osd_merge_lanes<32, 2>(kq, lane, [&](int) { return true; });
    // Blocks 2 (ascending) and 3 (all padding) merge DOWNWARDS: the padding moves to block 2, block 3 becomes block 2 reversed.
    uint64_t k3 = shfl64(kq[2], 63 - lane);
    // size 256, upwards.  Stride 128: (block 0, padding) stays; (block 1, block 3) exchange.  Stride 64: (block 0, block 1) exchange;
    // (padding, block 3) swap, i.e. block 2 := block 3.  Then the lane strides on blocks 0 .. 2.
    { const uint64_t a = kq[1], b = k3; const bool lt = a < b; kq[1] = lt ? a : b; k3 = lt ? b : a; }
    { const uint64_t a = kq[0], b = kq[1]; const bool lt = a < b; kq[0] = lt ? a : b; kq[1] = lt ? b : a; }
    kq[2] = k3;
    osd_merge_lanes<32, 3>(kq, lane, [&](int) { return true; });
#endif
    // ---- Gauss-Jordan over GF(2), generator held column-wise in SORTED order: lane l of register set s owns the column at
    // reliability position 64 s + l (91 row bits: rows 0..63 as two u32, rows 64..90 in a third).  A row is "locked" once it has
    // been made the unit row of an accepted column.  Visiting position ic (decoders.py:228-242 visits the columns in this order):
    // the column is broadcast to scalar registers (3 readlanes of a statically known register -- the loop is split per register
    // set); it is independent of the accepted columns iff it has a 1 in an unlocked row; its lowest such row r becomes the pivot
    // (which row is picked does not change the result: the reduced matrix of a given basis is unique up to row labels, and the
    // codeword / flip rows below are label-free); clearing the column's other 1s = adding row r to those rows = XORing (column
    // minus bit r) into every column that has a 1 in row r.  A column that already is a unit vector needs no update at all (the
    // still untouched systematic columns: about 40 % of the basis).  The scalar pipe issues one instruction per cycle per CU and
    // is this kernel's bottleneck (profiles/r02_notes.md), so the bookkeeping is kept to lock words and one accepted-position bit
    // per step; everything that can wait (hard-decision mask, flip rows, syndromes) is done afterwards on the vector side.
    OT(1);
    const int ord0 = (int)(uint32_t)kq[0], ord1 = (int)(uint32_t)kq[1], ord2 = (lane < 46) ? (int)(uint32_t)kq[2] : 0;
    const bool has2 = lane < 46;
    uint32_t x00 = d_G0T[ord0][0], x01 = d_G0T[ord0][1], x02 = d_G0T[ord0][2];
    uint32_t x10 = d_G0T[ord1][0], x11 = d_G0T[ord1][1], x12 = d_G0T[ord1][2];
    uint32_t x20 = has2 ? d_G0T[ord2][0] : 0u, x21 = has2 ? d_G0T[ord2][1] : 0u, x22 = has2 ? d_G0T[ord2][2] : 0u;
    // hard decisions: in natural order (distance test of the slow path) and per sorted position
    const uint64_t hard0 = __ballot(llr[lane] > 0.0f), hard1 = __ballot(llr[64 + lane] > 0.0f),
                   hard2 = __ballot(has2 && llr[128 + (has2 ? lane : 0)] > 0.0f);
    const bool hs0 = llr[ord0] > 0.0f, hs1 = llr[ord1] > 0.0f, hs2 = has2 && llr[ord2] > 0.0f;
    uint64_t lock01 = 0; uint32_t lock2 = ~((1u << 27) - 1u);
    uint64_t acc0 = 0, acc1 = 0, acc2 = 0;             // accepted positions per register set
    int k = 0;
#ifdef OSD_TIMING_SKIP_ELIM
    k = 91;
#endif
#define OSD_STEP(XA, XB, XC, ACC, IL)                                                                                              \
    {                                                                                                                              \
        const uint32_t c0 = __builtin_amdgcn_readlane(XA, IL), c1 = __builtin_amdgcn_readlane(XB, IL), c2 = __builtin_amdgcn_readlane(XC, IL); \
        const uint64_t c01 = ((uint64_t)c1 << 32) | c0;                                                                            \
        const uint64_t a01 = c01 & ~lock01; const uint32_t a2 = c2 & ~lock2;                                                       \
        if (a01 | a2) {                                        /* else: dependent on the accepted columns */                       \
            const uint64_t b01 = a01 & (0 - a01);              /* lowest unlocked row with a 1 */                                  \
            const uint32_t b2 = a01 ? 0u : (a2 & (0u - a2));                                                                       \
            const uint64_t m01 = c01 & ~b01; const uint32_t m2 = c2 & ~b2;                                                         \
            if (m01 | m2) {                                                                                                        \
                const uint32_t b0 = (uint32_t)b01, b1 = (uint32_t)(b01 >> 32), m0 = (uint32_t)m01, m1 = (uint32_t)(m01 >> 32);     \
                { const bool t = ((x00 & b0) | (x01 & b1) | (x02 & b2)) != 0; x00 ^= t ? m0 : 0u; x01 ^= t ? m1 : 0u; x02 ^= t ? m2 : 0u; } \
                { const bool t = ((x10 & b0) | (x11 & b1) | (x12 & b2)) != 0; x10 ^= t ? m0 : 0u; x11 ^= t ? m1 : 0u; x12 ^= t ? m2 : 0u; } \
                { const bool t = ((x20 & b0) | (x21 & b1) | (x22 & b2)) != 0; x20 ^= t ? m0 : 0u; x21 ^= t ? m1 : 0u; x22 ^= t ? m2 : 0u; } \
            }                                                                                                                      \
            lock01 |= b01; lock2 |= b2;                                                                                            \
            ACC |= 1ull << (IL);                                                                                                   \
            if (++k == 91) lim = 0;                                                                                                \
        }                                                                                                                          \
    }
    // one loop condition (il < lim; lim drops to 0 when the 91st column is accepted) and a 32-bit opaque counter: the two-condition
    // form cost 9 scalar instructions of loop control per step on the kernel's bottleneck pipe, this one 3
#define OSD_RUN(XA, XB, XC, ACC, N) { lim = (k < 91) ? (N) : 0; for (int il = 0; il < lim; il++) { asm volatile("" : "+s"(il)); OSD_STEP(XA, XB, XC, ACC, il) } }
    int lim;
    OT(2);
    OSD_RUN(x00, x01, x02, acc0, 64)
    OSD_RUN(x10, x11, x12, acc1, 64)
    OSD_RUN(x20, x21, x22, acc2, 46)
#undef OSD_RUN
#undef OSD_STEP
    OT(3);
    // Every accepted column is now a unit vector (its pivot row).  Acceptance order = position order, so the accepted column at
    // position p is the kk-th accepted one with kk = number of accepted positions before p.
    const bool in0 = (acc0 >> lane) & 1ull, in1 = (acc1 >> lane) & 1ull, in2 = (acc2 >> lane) & 1ull;
    const int n0 = __popcll(acc0), n1 = __popcll(acc1);
    const int kk0 = __builtin_amdgcn_mbcnt_hi((uint32_t)(acc0 >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)acc0, 0));
    const int kk1 = n0 + __builtin_amdgcn_mbcnt_hi((uint32_t)(acc1 >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)acc1, 0));
    const int kk2 = n0 + n1 + __builtin_amdgcn_mbcnt_hi((uint32_t)(acc2 >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)acc2, 0));
    // flip i = the row locked by accepted column 90 - i (least reliable basis members first): that column's lane publishes its unit vector
    if (lane < 3 * (OSD_MAXFLIP + 1)) frow[lane] = 0u;
    if (64 + lane < 3 * (OSD_MAXFLIP + 1)) frow[64 + lane] = 0u;
    if (128 + lane < 3 * (OSD_MAXFLIP + 1)) frow[128 + lane] = 0u;
    if (192 + lane < 3 * (OSD_MAXFLIP + 1)) frow[192 + lane] = 0u;
    if (256 + lane < 3 * (OSD_MAXFLIP + 1)) frow[256 + lane] = 0u;
    __syncthreads();
    { const int i = 90 - kk0; if (in0 && i >= 0 && i < nflip) { frow[3 * i] = x00; frow[3 * i + 1] = x01; frow[3 * i + 2] = x02; } }
    { const int i = 90 - kk1; if (in1 && i >= 0 && i < nflip) { frow[3 * i] = x10; frow[3 * i + 1] = x11; frow[3 * i + 2] = x12; } }
    { const int i = 90 - kk2; if (in2 && i >= 0 && i < nflip) { frow[3 * i] = x20; frow[3 * i + 1] = x21; frow[3 * i + 2] = x22; } }
    // hm = rows whose accepted column has hard decision 1 (OR of those unit vectors): order-0 codeword bit of a column = parity(column & hm)
    if (lane < 3) hmw[lane] = 0u;
    __syncthreads();
    {
        const uint32_t h0 = ((in0 && hs0) ? x00 : 0u) | ((in1 && hs1) ? x10 : 0u) | ((in2 && hs2) ? x20 : 0u);
        const uint32_t h1 = ((in0 && hs0) ? x01 : 0u) | ((in1 && hs1) ? x11 : 0u) | ((in2 && hs2) ? x21 : 0u);
        const uint32_t h2 = ((in0 && hs0) ? x02 : 0u) | ((in1 && hs1) ? x12 : 0u) | ((in2 && hs2) ? x22 : 0u);
        if (h0) atomicOr(&hmw[0], h0);
        if (h1) atomicOr(&hmw[1], h1);
        if (h2) atomicOr(&hmw[2], h2);
    }
    __syncthreads();
    const uint32_t hm0 = hmw[0], hm1 = hmw[1], hm2 = hmw[2];
    OT(4);
    // per column: bit i = flip i has a 1 in this column (i < 62), bit 63 = the order-0 codeword bit
    uint64_t f0 = (uint64_t)((__popc(x00 & hm0) + __popc(x01 & hm1) + __popc(x02 & hm2)) & 1) << 63,
             f1 = (uint64_t)((__popc(x10 & hm0) + __popc(x11 & hm1) + __popc(x12 & hm2)) & 1) << 63,
             f2 = (uint64_t)((__popc(x20 & hm0) + __popc(x21 & hm1) + __popc(x22 & hm2)) & 1) << 63;
    for (int i = 0; i < nflipA; i++) {
        const uint32_t r0 = frow[3 * i], r1 = frow[3 * i + 1], r2 = frow[3 * i + 2];      // broadcast reads (a unit vector, or 0 if absent)
        f0 |= (uint64_t)(((x00 & r0) | (x01 & r1) | (x02 & r2)) != 0) << i;
        f1 |= (uint64_t)(((x10 & r0) | (x11 & r1) | (x12 & r2)) != 0) << i;
        f2 |= (uint64_t)(((x20 & r0) | (x21 & r1) | (x22 & r2)) != 0) << i;
    }
    // back to natural column order: ftab[column]
    ftab[ord0] = f0; ftab[ord1] = f1; if (has2) ftab[ord2] = f2;
    if (wide) {                                            // flips 62 .. nflip - 1 into the second word
        uint32_t g0 = 0, g1 = 0, g2 = 0;
        for (int i = OSD_FLIPS_A; i < nflip; i++) {
            const uint32_t r0 = frow[3 * i], r1 = frow[3 * i + 1], r2 = frow[3 * i + 2];
            g0 |= (uint32_t)(((x00 & r0) | (x01 & r1) | (x02 & r2)) != 0) << (i - OSD_FLIPS_A);
            g1 |= (uint32_t)(((x10 & r0) | (x11 & r1) | (x12 & r2)) != 0) << (i - OSD_FLIPS_A);
            g2 |= (uint32_t)(((x20 & r0) | (x21 & r1) | (x22 & r2)) != 0) << (i - OSD_FLIPS_A);
        }
        ftabB[ord0] = g0; ftabB[ord1] = g1; if (has2) ftabB[ord2] = g2;
    }
    __syncthreads();
    OT(5);
    // CRC syndromes (the CRC is linear): lane i < min(nflip, 62) takes flip i, lane 63 the order-0 codeword (flips 62.. in a second round).  The lane gathers its word as a
    // 91-bit column set (bit `bitsel` of every ftab entry), then each of the 14 syndrome bits is a masked parity (d_SYNM)
    {
        const int bitsel = (lane < nflipA) ? lane : 63;
        const bool up = bitsel >= 32;
        const int sh = bitsel & 31;
        uint32_t ra = 0, rb = 0, rc = 0;
#pragma unroll 8
        for (int v = 0; v < 32; v++) {
            const uint64_t a = ftab[v], b = ftab[32 + v], c = ftab[64 + (v < 27 ? v : 0)];
            ra |= (((up ? (uint32_t)(a >> 32) : (uint32_t)a) >> sh) & 1u) << v;
            rb |= (((up ? (uint32_t)(b >> 32) : (uint32_t)b) >> sh) & 1u) << v;
            rc |= (((up ? (uint32_t)(c >> 32) : (uint32_t)c) >> sh) & 1u) << v;
        }
        rc &= (1u << 27) - 1;
        unsigned sy = 0;
#pragma unroll
        for (int k = 0; k < 14; k++)
            sy |= (unsigned)((__popc(ra & d_SYNM[k][0]) + __popc(rb & d_SYNM[k][1]) + __popc(rc & d_SYNM[k][2])) & 1) << k;
        if (lane < nflipA) fsyn[lane] = (uint16_t)sy;
        if (lane == 63) fsyn[OSD_MAXFLIP + 1] = (uint16_t)sy;
        if (lane == 0) fsyn[OSD_MAXFLIP] = 0;
    }
    if (wide) {                                            // lane l takes flip 62 + l: bit l of the second word of every column
        const int sh = lane & 31;
        uint32_t ra = 0, rb = 0, rc = 0;
#pragma unroll 8
        for (int v = 0; v < 32; v++) {
            ra |= ((ftabB[v] >> sh) & 1u) << v;
            rb |= ((ftabB[32 + v] >> sh) & 1u) << v;
            rc |= ((ftabB[64 + (v < 27 ? v : 0)] >> sh) & 1u) << v;
        }
        rc &= (1u << 27) - 1;
        unsigned sy = 0;
#pragma unroll
        for (int k = 0; k < 14; k++)
            sy |= (unsigned)((__popc(ra & d_SYNM[k][0]) + __popc(rb & d_SYNM[k][1]) + __popc(rc & d_SYNM[k][2])) & 1) << k;
        if (OSD_FLIPS_A + lane < nflip) fsyn[OSD_FLIPS_A + lane] = (uint16_t)sy;
    }
    __syncthreads();
    OT(6);
    const unsigned syn_c = fsyn[OSD_MAXFLIP + 1];
    const uint64_t M1 = (1ull << 27) - 1, M2 = (1ull << 46) - 1;
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
        uint64_t hits = __ballot(hit);
        if (!hits) continue;                                  // no CRC-consistent word among these 64 trials (the usual case)
        // slow path, in trial order: rebuild the candidate codeword (natural column order) from the per-column flip words, apply the
        // distance gate, run the validity predicate, log the reference's unpack() call; the first accepted trial wins
        bool done = false;
        while (hits && !done) {
            const int hl = __builtin_ctzll(hits);
            hits &= hits - 1;
            const int hi_ = __shfl(i, hl), hj = __shfl(j, hl), hq = __shfl(q, hl);
            const uint64_t msk = (1ull << 63) | ((hi_ < OSD_FLIPS_A) ? (1ull << hi_) : 0ull) | ((hj < OSD_FLIPS_A) ? (1ull << hj) : 0ull) |
                                 ((hq < OSD_FLIPS_A) ? (1ull << hq) : 0ull);
            uint32_t mskB = 0;                                // flips 62..90 (the "no flip" index 91 sets nothing)
            if (wide) mskB = ((hi_ >= OSD_FLIPS_A && hi_ < OSD_MAXFLIP) ? (1u << (hi_ - OSD_FLIPS_A)) : 0u) |
                             ((hj >= OSD_FLIPS_A && hj < OSD_MAXFLIP) ? (1u << (hj - OSD_FLIPS_A)) : 0u) |
                             ((hq >= OSD_FLIPS_A && hq < OSD_MAXFLIP) ? (1u << (hq - OSD_FLIPS_A)) : 0u);
            const int l2 = 128 + (has2 ? lane : 0);
            const int pb0 = wide ? __popc(ftabB[lane] & mskB) : 0, pb1 = wide ? __popc(ftabB[64 + lane] & mskB) : 0, pb2 = wide ? __popc(ftabB[l2] & mskB) : 0;
            const uint64_t w0 = __ballot((__popcll(ftab[lane] & msk) + pb0) & 1), w1 = __ballot((__popcll(ftab[64 + lane] & msk) + pb1) & 1),
                           w2 = __ballot(has2 && ((__popcll(ftab[l2] & msk) + pb2) & 1));
            const int hd = __popcll(w0 ^ hard0) + __popcll(w1 ^ hard1) + __popcll((w2 ^ hard2) & M2);
            if (max_hd > 0 && hd > max_hd) continue;          // gate (extension): no unpack() call beyond max_hd
            uint64_t lo = 0, hi = 0;
            const int r = ft8_crc_check(w0, w1 & M1, &lo, &hi);
            const int t = base + hl;
            if (r && lane == 0) log_event(ev, evcount, frame, ci, ipass, slot, t, lo, hi, r == 2);   // a call the reference made
            if (r == 2) {
                res.ok = 1; res.lo = lo; res.hi = hi; res.n_its = (int16_t)t;
                res.method = (slot < 5) ? FT8RX_M_OSD : FT8RX_M_LDPC_B_OSD;
                res.pad[0] = (uint8_t)hd;                     // Hamming distance of the accepted codeword to the hard decisions
                done = true;
            }
        }
        if (done) break;
    }
    OT(7);
    if (lane == 0) attO[vec] = res;
    OT_FLUSH;
}


// mode 2 (test entry): one block per vector.  Pipeline: blocks stride over OSD work list x 10 attempts (5 AP variants of the fine
// LLRs, then the 5 saved BP outputs).
// Eight waves per SIMD: the elimination is a serial chain of readlane -> scalar logic -> masked XOR per step (~280 cycles), hidden only
// by other waves.  The kernel needs 59 VGPRs when told to fit eight (71 otherwise) and 4.5 KB of LDS since the flip rows share the
// dead sort keys (5.6 KB allowed 7): 0.739 -> 0.710 ms per 256 frames (profiles/r03_notes.md; the attribute alone, LDS-bound at 7: 0.781)
#ifndef OSD_ATTR
#define OSD_ATTR __attribute__((amdgpu_waves_per_eu(8, 8)))
#endif
#define OSD_KERNEL(NAME, WIDE)                                                                                                       \
__global__ __launch_bounds__(64) OSD_ATTR void NAME(int mode, const float* __restrict__ llr_in, const float* __restrict__ saved,             \
                                           const Att* __restrict__ attB, ft8rx_record* __restrict__ rec,                            \
                                           const int32_t* __restrict__ ncand, Att* __restrict__ attO,                               \
                                           ft8rx_event* ev, int32_t* evcount, const uint32_t* __restrict__ trials, int ntr,        \
                                           int nflip, int max_hd, WorkList work) {                                                  \
    if (mode == 2) { osd_attempt<WIDE>(threadIdx.x, 2, blockIdx.x, llr_in, saved, attB, rec, ncand, attO, ev, evcount, trials, ntr, nflip, max_hd); return; } \
    const int n = *work.count * 10;                                                                                                 \
    _Pragma("unroll 1")                                                                                                             \
    for (int item = blockIdx.x; item < n; item += gridDim.x) {                                                                      \
        int lane = threadIdx.x;                                                                                                     \
        asm volatile("" : "+v"(lane));       /* opaque per item: nothing lane-specific is hoisted across attempts (register pressure) */ \
        osd_attempt<WIDE>(lane, 0, work.items[item / 10] * 10 + item % 10, llr_in, saved, attB, rec, ncand, attO, ev, evcount, trials, ntr, nflip, max_hd); \
        __syncthreads();                     /* the LDS arrays are reused by the next attempt */                                   \
    }                                                                                                                               \
}
OSD_KERNEL(k_osd, false)
OSD_KERNEL(k_osd_wide, true)        /* more than OSD_FLIPS_A flip rows */
#undef OSD_KERNEL

#endif
