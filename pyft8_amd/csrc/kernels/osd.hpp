// osd.hpp -- ordered-statistics decoding (decoders.py:223-272)
// Part of libft8rx.so; included by ft8rx.hip (single translation unit: the kernels share __constant__/__device__ tables).
#ifndef FT8RX_OSD_HPP
#define FT8RX_OSD_HPP

// ------------------------------------------------------------------------------------ OSD (decoders.py:223-272)
// One wavefront per attempt, three phases:
//   1. reliability order = np.argsort(-abs(llr)) as the reference's numpy (2.2.6 on AVX-512: x86-simd-sort) orders it, equal keys
//      included: that library's 256-wire compare-exchange network run on (key, index) pairs held in registers, exchanged by DPP /
//      ds_swizzle / ds_bpermute; a vector with a NaN takes the library's other path, std::sort, on one lane (rare);
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
#ifdef OSD_COUNT_VISITS
#define OT_VISIT do { if (lane == 0) g_osd_t[blockIdx.x & 32767][8] += 1ull; } while (0)      /* one visited column (its own build: the store costs every visit) */
#else
#define OT_VISIT do { } while (0)
#endif
#else
#define OT_DECL
#define OT(i) do { } while (0)
#define OT_FLUSH do { } while (0)
#define OT_VISIT do { } while (0)
#endif
// ---- np.argsort's compare-exchange network (oracle/ft8_oracle.c: ft8o_argsort_f32 has the derivation and the library references).
// 256 wires = 32 registers of 8 lanes in the library; here wire w = 64 q + lane sits in register q of the lane.  Every stage pairs wire w
// with w ^ M, the LOWER wire keeps the smaller key, and equal keys never move (the library moves an index only when min / max did not
// return the lane's own key).  Stage masks: 1 3 1 7 2 1 | 15 4 2 1 | 31 8 4 2 1 | 63 16 8 4 2 1 | 127 32 16 8 4 2 1 | 255 64 32 16 8 4 2 1.
// osd_px<M>: the value of lane ^ M, M < 64: quad permutes / row mirrors (DPP) for 1, 2, 3, 7, 15, a row rotate for 8, ds_swizzle (crossbar only,
// no address register) for 4, 16, 31, ds_bpermute for 32 and 63.
template <int M> FT8_DEV uint32_t osd_px(uint32_t v, int lane) {
    if (M == 1) return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1, 0xf, 0xf, false);          // quad_perm:[1,0,3,2]
    if (M == 2) return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x4E, 0xf, 0xf, false);          // quad_perm:[2,3,0,1]
    if (M == 3) return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x1B, 0xf, 0xf, false);          // quad_perm:[3,2,1,0]
    if (M == 7) return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x141, 0xf, 0xf, false);         // row_half_mirror
    if (M == 15) return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x140, 0xf, 0xf, false);        // row_mirror
    if (M == 8) return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x128, 0xf, 0xf, false);         // row_ror:8
    if (M == 4 || M == 16 || M == 31) return (uint32_t)__builtin_amdgcn_ds_swizzle((int)v, 0x1f | (M << 10));   // bit mode: lane ^ M within 32
    return __shfl(v, lane ^ M);
}
// One stage between lanes M apart on NQ registers of (key, index) pairs.  The sort is a fifth of the attempt's vector instructions and the
// kernel is bound by their number, so a register-stage is written out: both orders of the pair (min, max), the lane's side picked with a
// constant lane mask (a scalar operand), "the key changed" = "take the partner's index" -- 5 vector instructions where the compare-and-
// select form compiled to ~14.  For the strides a DPP control reaches (1, 2, 3, 7, 15, 8) the partner is read as a DPP operand of v_min /
// v_max / v_cndmask themselves (inline assembly: the compiler only fuses a DPP move into a single user).  Hazards the assembler does not
// see inside an asm block: VALU writes VCC -> VALU reads VCC needs two wait states on gfx950 (the s_nop below); a DPP read of a VGPR needs
// two wait states after the VALU write -- the blocks of the other register sets (or of the previous stage) lie in between.
#define OSD_LOWMASK(top) ((top) == 1 ? 0x5555555555555555ull : (top) == 2 ? 0x3333333333333333ull : (top) == 4 ? 0x0F0F0F0F0F0F0F0Full : \
                          (top) == 8 ? 0x00FF00FF00FF00FFull : (top) == 16 ? 0x0000FFFF0000FFFFull : 0x00000000FFFFFFFFull)
#define OSD_DPP_STAGE(CTRL)                                                                                                          \
    asm volatile("v_min_u32_dpp %[mn], %[hk], %[hk] " CTRL " row_mask:0xf bank_mask:0xf\n\t"                                         \
                 "v_max_u32_dpp %[mx], %[hk], %[hk] " CTRL " row_mask:0xf bank_mask:0xf\n\t"                                         \
                 "v_cndmask_b32 %[mn], %[mx], %[mn], %[low]\n\t"                                                                    \
                 "v_cmp_eq_u32 vcc, %[mn], %[hk]\n\t"                                                                                \
                 "s_nop 1\n\t"                                                                                                       \
                 "v_cndmask_b32_dpp %[ix], %[ix], %[ix], vcc " CTRL " row_mask:0xf bank_mask:0xf\n\t"                                \
                 : [mn] "=&v"(mn), [mx] "=&v"(mx), [ix] "+v"(ix[q]) : [hk] "v"(hk[q]), [low] "s"(low) : "vcc");
template <int M, int NQ> FT8_DEV void osd_stage(uint32_t* hk, uint32_t* ix, int lane) {
    constexpr int top = M >= 32 ? 32 : M >= 16 ? 16 : M >= 8 ? 8 : M >= 4 ? 4 : M >= 2 ? 2 : 1;      // highest bit of M: clear on the lower wire
    const uint64_t low = OSD_LOWMASK(top);                     // lanes that hold the lower wire of their pair
#pragma unroll
    for (int q = 0; q < NQ; q++) {
        uint32_t mn, mx;
        if constexpr (M == 1) { OSD_DPP_STAGE("quad_perm:[1,0,3,2]") }
        else if constexpr (M == 2) { OSD_DPP_STAGE("quad_perm:[2,3,0,1]") }
        else if constexpr (M == 3) { OSD_DPP_STAGE("quad_perm:[3,2,1,0]") }
        else if constexpr (M == 7) { OSD_DPP_STAGE("row_half_mirror") }
        else if constexpr (M == 15) { OSD_DPP_STAGE("row_mirror") }
        else if constexpr (M == 8) { OSD_DPP_STAGE("row_ror:8") }
        else {
            const uint32_t pk = osd_px<M>(hk[q], lane), pi = osd_px<M>(ix[q], lane);
            mn = pk < hk[q] ? pk : hk[q]; mx = pk < hk[q] ? hk[q] : pk;
            asm("v_cndmask_b32 %0, %1, %0, %2" : "+v"(mn) : "v"(mx), "s"(low));      // lower wire: the minimum
            ix[q] = (mn == hk[q]) ? ix[q] : pi;
        }
        hk[q] = mn;
    }
}
#undef OSD_DPP_STAGE
template <int NQ> FT8_DEV void osd_stages_421(uint32_t* hk, uint32_t* ix, int lane) {
    osd_stage<4, NQ>(hk, ix, lane); osd_stage<2, NQ>(hk, ix, lane); osd_stage<1, NQ>(hk, ix, lane);
}
// register a (lower wires) against register b, same lane (reversed = false) or lane 63 - l (reversed = true: the first stage of a merge)
template <bool REV> FT8_DEV void osd_stage_regs(uint32_t& ka, uint32_t& ia, uint32_t& kb, uint32_t& ib, int lane) {
    const uint32_t pkb = REV ? __shfl(kb, 63 - lane) : kb, pib = REV ? __shfl(ib, 63 - lane) : ib;      // what a's lane faces
    const uint32_t pka = REV ? __shfl(ka, 63 - lane) : ka, pia = REV ? __shfl(ia, 63 - lane) : ia;      // what b's lane faces
    const uint32_t na = pkb < ka ? pkb : ka, nb = pka > kb ? pka : kb;                                   // a keeps the minimum, b the maximum
    ia = (na == ka) ? ia : pib; ib = (nb == kb) ? ib : pia;
    ka = na; kb = nb;
}
// ---- the library's path for a vector that contains a NaN (std_argsort_withnan): libstdc++'s std::sort of the index array with the
// comparator "both not NaN: a < b; a NaN: false; else true" on the keys -|llr|.  One lane, in LDS; rare (a NaN-poisoned BP output,
// decoders.py:143-147).  Function for function as in oracle/ft8_oracle.c (std_sort_withnan), with the recursion on the right part
// turned into an explicit stack (the parts are disjoint, their order does not matter).
#define FT8_HD __host__ __device__ inline
FT8_HD bool osd_nl(const float* x, int a, int b) {
    const float xa = x[a], xb = x[b];
    if (xa == xa && xb == xb) return fabsf(xa) > fabsf(xb);             // -|xa| < -|xb|
    return xa == xa;                                                    // a NaN: false; a number against a NaN: true
}
FT8_HD void osd_ss_linear_insert(const float* x, int* a, int last) {
    const int val = a[last]; int next = last - 1;
    while (osd_nl(x, val, a[next])) { a[last] = a[next]; last = next; next--; }
    a[last] = val;
}
FT8_HD void osd_ss_insertion(const float* x, int* a, int first, int last) {
    for (int i = first + 1; i < last; i++) {
        if (osd_nl(x, a[i], a[first])) { const int val = a[i]; for (int j = i; j > first; j--) a[j] = a[j - 1]; a[first] = val; }
        else osd_ss_linear_insert(x, a, i);
    }
}
FT8_HD void osd_ss_adjust_heap(const float* x, int* a, int first, int hole, int len, int value) {
    const int top = hole; int child = hole;
    while (child < (len - 1) / 2) {
        child = 2 * (child + 1);
        if (osd_nl(x, a[first + child], a[first + child - 1])) child--;
        a[first + hole] = a[first + child]; hole = child;
    }
    if ((len & 1) == 0 && child == (len - 2) / 2) { child = 2 * (child + 1); a[first + hole] = a[first + child - 1]; hole = child - 1; }
    int parent = (hole - 1) / 2;                                        // __push_heap
    while (hole > top && osd_nl(x, a[first + parent], value)) { a[first + hole] = a[first + parent]; hole = parent; parent = (hole - 1) / 2; }
    a[first + hole] = value;
}
FT8_HD void osd_std_sort_withnan(const float* x, int* a, int* stack /* [3 * 20] */) {
    const int n = 174;
    for (int i = 0; i < n; i++) a[i] = i;
    int sp = 0;
    stack[0] = 0; stack[1] = n; stack[2] = 2 * 7; sp = 1;              // depth limit 2 * floor(lg 174)
    while (sp > 0) {
        sp--;
        int first = stack[3 * sp], last = stack[3 * sp + 1], depth = stack[3 * sp + 2];
        while (last - first > 16) {
            if (depth == 0) {                                           // heapsort (__partial_sort(first, last, last))
                const int len = last - first;
                for (int parent = (len - 2) / 2; parent >= 0; parent--) osd_ss_adjust_heap(x, a, first, parent, len, a[first + parent]);
                while (last - first > 1) { last--; const int v = a[last]; a[last] = a[first]; osd_ss_adjust_heap(x, a, first, 0, last - first, v); }
                break;
            }
            depth--;
            {   // __move_median_to_first(first, first + 1, mid, last - 1)
                const int p = first + 1, q = first + (last - first) / 2, r = last - 1;
                int pick;
                if (osd_nl(x, a[p], a[q])) pick = osd_nl(x, a[q], a[r]) ? q : (osd_nl(x, a[p], a[r]) ? r : p);
                else pick = osd_nl(x, a[p], a[r]) ? p : (osd_nl(x, a[q], a[r]) ? r : q);
                const int t = a[first]; a[first] = a[pick]; a[pick] = t;
            }
            int lo = first + 1, hi = last;                              // __unguarded_partition(first + 1, last, pivot = *first)
            for (;;) {
                while (osd_nl(x, a[lo], a[first])) lo++;
                hi--;
                while (osd_nl(x, a[first], a[hi])) hi--;
                if (!(lo < hi)) break;
                const int t = a[lo]; a[lo] = a[hi]; a[hi] = t;
                lo++;
            }
            stack[3 * sp] = lo; stack[3 * sp + 1] = last; stack[3 * sp + 2] = depth; sp++;      // the right part, later
            last = lo;
        }
    }
    osd_ss_insertion(x, a, 0, 16);                                      // __final_insertion_sort, n > 16
    for (int i = 16; i < n; i++) osd_ss_linear_insert(x, a, i);
}

#undef FT8_HD
// the order std::sort leaves an ALL-NaN vector in (the comparator is then always false: a fixed permutation; ft8rx_create fills it by
// running the routine above on the host): the common NaN case -- a NaN-poisoned BP output is NaN everywhere -- stays in the main kernel
__device__ uint8_t d_NANPERM[192];

#define OSD_MAXFLIP 91            /* flip rows kept per attempt = all 91 basis positions (decoders.py:244-246 takes any count up to the basis size) */
#define OSD_FLIPS_A 62            /* flips 0..61: one bit each in the column's 64-bit word (bit 63 = order-0 codeword bit); flips 62..90: a second, 32-bit word
                                     that only exists for attempts with more than 62 flip rows (the reference's own callers use 30 / 40) */
#define OSD_MAXTRIALS 16384       /* trial index must fit the 16-bit seq of the event log */
#define OSD_NONE 0xFFu            /* "no flip" in a packed trial entry (i | j << 8 | k << 16) */

// CRC syndrome of codeword bit v alone (v < 77: message bit, 77..90: the CRC field bit itself), bit-sliced: d_SYNM[k][w] bit b = bit k
// of the syndrome of codeword bit 32 w + b.  Constant address space: wave-uniform reads become scalar loads (a uniform lookup in a
// 16-bit __device__ table is a broadcast vector load -- 91 of them per attempt kept the texture-address path busy, profiles/archive/r02_notes.md)
__device__ __constant__ uint32_t d_SYNM[14][3];
__device__ uint32_t d_G0T[192][3];       // column v of G0 = [I | A^T]: row bits 0..31, 32..63, 64..90 (columns >= 174 are zero)
FT8_DEV unsigned osd_syndrome(uint64_t w0, uint64_t w1) { return ft8_crc_syndrome(w0, w1); }     // table d_CRC_T: ft8_dev.h

// mode 0: pipeline (work = (candidate, slot 0..9)); mode 2: raw vectors.  WIDE: more than 62 flip rows (k_osd_wide) -- a kernel of its
// own, because the second flip word costs 13 VGPRs = two of the seven waves per SIMD that hide this kernel's scalar-pipe latency
// (one kernel with a run-time switch: 0.737 -> 0.791 ms per 256 frames at the reference's 30 / 2)
// NANV: the kernel of the attempts whose vector holds a NaN (k_osd_nan / k_osd_nan_wide).  The library sorts such a vector with std::sort
// -- a serial algorithm, run by one lane, whose code costs the main kernel 8 VGPRs and a scratch frame if it lives there: the main kernels
// (NANV = false) only append such an attempt to `nanlist` and leave; the NaN kernels stride over that list (almost always empty).
template <bool WIDE, bool NANV>
FT8_DEV void osd_attempt(int lane, int mode, int bid, const float* __restrict__ llr_in, const float* __restrict__ saved,
                         const Att* __restrict__ attB, ft8rx_record* __restrict__ rec,
                         const int32_t* __restrict__ ncand, Att* __restrict__ attO,
                         ft8rx_event* ev, int32_t* evcount, const uint32_t* __restrict__ trials, int ntr,
                         int nflip, int max_hd, const WorkList& nanlist) {
    __shared__ float llr[176];
    __shared__ uint64_t skey[256];
    __shared__ uint64_t ftab[192];                         // per column (natural order): bit i = flip i covers it (i < 62), bit 63 = order-0 codeword bit
    // the sort keys are dead once the reliability order has been read into registers; their 2 KB then hold
    uint32_t* ftabB = reinterpret_cast<uint32_t*>(skey);  // [192] bit i - 62 = flip i covers it (62 <= i < 91), and
    uint32_t* frow = ftabB + 192;                          // [OSD_MAXFLIP] pivot rows of the flip columns (row indices)
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
    // ---- reliability order: np.argsort(-abs(llr)) as the reference's numpy orders it (decoders.py:226; the network above).  Keys: the
    // magnitude bits, inverted so that an ascending unsigned sort is |llr| descending; padding wires (174 .. 255) hold the largest key.
    // Register 3 would be all padding and stays all padding up to the last merge: every stage that involves it is resolved by hand.
    uint32_t hk[3], ix[3];
#pragma unroll
    for (int q = 0; q < 3; q++) {
        const int i = lane + 64 * q;
        const uint32_t mag = __float_as_uint(llr[i < 174 ? i : 0]) & 0x7fffffffu;
        hk[q] = (i < 174) ? 0xFFFFFFFEu - mag : 0xFFFFFFFFu;
        ix[q] = (i < 174) ? (uint32_t)i : 0u;
    }
    const uint64_t nan0 = __ballot(hk[0] < 0x807FFFFEu), nan1 = __ballot(hk[1] < 0x807FFFFEu), nan2 = __ballot(lane < 46 && hk[2] < 0x807FFFFEu);      // magnitude bits above infinity's
    const bool any_nan = (nan0 | nan1 | nan2) != 0, all_nan = (nan0 & nan1) == ~0ull && nan2 == (1ull << 46) - 1;
    OT(0);
#ifndef OSD_TIMING_SKIP_SORT            /* timing-only builds (tools/ab_variants.sh): never defined in the product */
    if (NANV) {                                               // the library's std::sort path, one lane (k_osd_nan)
        if (!any_nan || all_nan) return;                      // (never: the list only holds vectors with some, not only, NaNs)
        int* ordl = reinterpret_cast<int*>(skey);             // [174] the order, then [60] the range stack (skey is not in use yet)
        if (lane == 0) osd_std_sort_withnan(llr, ordl, ordl + 176);
        __syncthreads();
        ix[0] = (uint32_t)ordl[lane]; ix[1] = (uint32_t)ordl[64 + lane]; ix[2] = (uint32_t)ordl[128 + (lane < 46 ? lane : 0)];
        __syncthreads();
    } else if (all_nan) {                                     // std::sort of 174 NaNs: a fixed permutation
        ix[0] = d_NANPERM[lane]; ix[1] = d_NANPERM[64 + lane]; ix[2] = d_NANPERM[128 + lane];
    } else if (any_nan) {                                     // wave-uniform, rare: left to the NaN kernel
        if (lane == 0) work_push(nanlist, bid);
        return;
    } else {
        // inside the registers: each register's 64 wires sorted (8 library registers), three at a time
        osd_stage<1, 3>(hk, ix, lane); osd_stage<3, 3>(hk, ix, lane); osd_stage<1, 3>(hk, ix, lane);
        osd_stage<7, 3>(hk, ix, lane); osd_stage<2, 3>(hk, ix, lane); osd_stage<1, 3>(hk, ix, lane);
        osd_stage<15, 3>(hk, ix, lane); osd_stages_421<3>(hk, ix, lane);
        osd_stage<31, 3>(hk, ix, lane); osd_stage<8, 3>(hk, ix, lane); osd_stages_421<3>(hk, ix, lane);
        osd_stage<63, 3>(hk, ix, lane); osd_stage<16, 3>(hk, ix, lane); osd_stage<8, 3>(hk, ix, lane); osd_stages_421<3>(hk, ix, lane);
        // merge of 16 library registers: wires 0..63 against 127..64 reversed (registers 0 / 1); registers 2 / 3: 3 is padding, 2 is
        // sorted and stays as it is through this whole merge
        osd_stage_regs<true>(hk[0], ix[0], hk[1], ix[1], lane);
        osd_stage<32, 2>(hk, ix, lane); osd_stage<16, 2>(hk, ix, lane); osd_stage<8, 2>(hk, ix, lane); osd_stages_421<2>(hk, ix, lane);
        // merge of 32: w against 255 - w (register 1 against register 2 reversed; register 0 faces padding), then registers 0 / 1 lane
        // by lane (2 faces padding), then the lane strides on all three
        osd_stage_regs<true>(hk[1], ix[1], hk[2], ix[2], lane);
        osd_stage_regs<false>(hk[0], ix[0], hk[1], ix[1], lane);
        osd_stage<32, 3>(hk, ix, lane); osd_stage<16, 3>(hk, ix, lane); osd_stage<8, 3>(hk, ix, lane); osd_stages_421<3>(hk, ix, lane);
    }
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
    // is this kernel's bottleneck (profiles/archive/r02_notes.md), so the bookkeeping is kept to lock words and one accepted-position bit
    // per step; everything that can wait (hard-decision mask, flip rows, syndromes) is done afterwards on the vector side.
    OT(1);
    const int ord0 = (int)ix[0], ord1 = (int)ix[1], ord2 = (lane < 46) ? (int)ix[2] : 0;
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
#ifndef OSD_VISIT_ALL                /* -DOSD_VISIT_ALL: the round-4 loop (every position visited), for the A/B of tools/ab_variants.sh */
    // SYSTEMATIC COLUMNS ARE NOT VISITED (round 5).  Column v < 91 of G0 = [I | A^T] is the unit vector e_v, and it stays e_v until some
    // pivot takes row v (an elimination only touches columns with a 1 in the pivot row).  Reached with row v free it is accepted with
    // pivot v and changes nothing -- about half of the ~105 visited positions, each costing the full broadcast / test / lock round on
    // the scalar pipe, the kernel's bottleneck.  So: the systematic columns at positions < OSD_TRIV ("trivial") are accepted without a
    // visit, the loop walks the other positions only (a bit mask per register set, find-first-set), and the free choice of the pivot
    // row keeps them trivial: a visited column takes its pivot among the rows that do NOT belong to a trivial column (lockU = locked or
    // reserved) -- there are as many such rows as pivots needed, up to the few positions the basis ends before or after OSD_TRIV.
    // Only a column with no 1 left there STEALS: among its 1s in rows of trivial columns positioned AFTER it (a trivial column before
    // it has been accepted: its row is locked) it takes the lowest, and the robbed column -- no longer a unit vector -- goes back
    // onto the visit list.  The basis is complete when visited accepts + trivial positions passed reach 91 (tcN: the number of
    // trivial positions before a lane).  Same information set as the plain loop, position for position (tests: info set == oracle's).
#define OSD_TRIV 96
    int* posrow = reinterpret_cast<int*>(skey);               // [96] position of systematic column v (skey is idle until the flip rows)
    if (ord0 < 91) posrow[ord0] = lane;
    if (ord1 < 91) posrow[ord1] = 64 + lane;
    if (has2 && ord2 < 91) posrow[ord2] = 128 + lane;
    uint64_t triv0 = __ballot(ord0 < 91), triv1 = __ballot(ord1 < 91 && lane < OSD_TRIV - 64);
    int tc0 = __builtin_amdgcn_mbcnt_hi((uint32_t)(triv0 >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)triv0, 0));
    int tc1 = __popcll(triv0) + __builtin_amdgcn_mbcnt_hi((uint32_t)(triv1 >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)triv1, 0));
    int ntriv = __popcll(triv0) + __popcll(triv1);
    __syncthreads();
    // rows reserved for trivial columns: bit r = systematic column r sits at a position < OSD_TRIV
    const uint64_t u01 = __ballot(posrow[lane] < OSD_TRIV);
    const uint32_t u2 = (uint32_t)__ballot(lane < 27 && posrow[64 + (lane < 27 ? lane : 0)] < OSD_TRIV);
    uint64_t lockU01 = u01; uint32_t lockU2 = lock2 | u2;      // locked by a pivot, or reserved for a trivial column
    uint64_t vis0 = ~triv0, vis1 = ~triv1, vis2 = (1ull << 46) - 1;
    // One visit = one straight path, integers only on the scalar side (a boolean that lives across a branch becomes a lane mask and
    // drags selects onto the vector pipe).  Both pipes are close to their ceilings in this loop -- eight waves x (SALU + VALU per visit)
    // x 4 cycles is what a round of visits takes (profiles/r05_notes.md) -- so the visit is counted in instructions:
    //   * every visit marks its position accepted and counts it (ACC, k) without a test; the rare paths take that back: the steal
    //     branch only replaces the eligible rows (a01, a2), and a dependent column (no eligible row at all) clears its mark, uncounts
    //     itself and runs the rest with an empty pivot b = 0 -- no column "has a 1 in row b", the update mask is zero everywhere;
    //   * the locked rows exist once (lockU = pivots taken + rows reserved for trivial columns); the steal branch works out which
    //     reserved rows are up for grabs from posrow and the trivial masks;
    //   * the update: "a 1 in the pivot row" as an all-ones / zero mask, then (mask & m) ^ x per word -- one v_bitop3 each on gfx950;
    //   * columns at positions already passed never change again (an accepted one is a unit vector at a locked row, a rejected one has
    //     all its 1s in rows that were locked when it was visited, and a pivot row is an unlocked row): while register set 1 is
    //     walked, set 0 is left alone; while set 2 is walked, sets 0 and 1.
    // TCV: trivial positions before a lane of this register set; BASE: its first position; VIS: its visit mask.  When the basis is
    // complete every visit mask is emptied, which ends this loop and skips the ones that follow.
    // (the accept step exists three times, once per word the pivot row can lie in -- the lowest eligible row is looked for word by word
    // anyway -- so that "a 1 in the pivot row" is ONE v_bfe_i32 of a statically known register: 4 vector instructions per register set)
#define OSD_UPD(XW, X0, X1, X2) { const uint32_t mk = (uint32_t)__builtin_amdgcn_sbfe((int)(XW), bp, 1u); X0 ^= mk & m0; X1 ^= mk & m1; X2 ^= mk & m2; }
#define OSD_STEP(XA, XB, XC, ACC, VIS, TCV, BASE, UPD0, UPD1, UPD2)                                                                \
    while (VIS != 0) {                                                                                                             \
        const int il = __builtin_ctzll(VIS);                                                                                       \
        OT_VISIT;                                                                                                                  \
        if (k + __builtin_amdgcn_readlane(TCV, il) >= 91) { vis0 = 0; vis1 = 0; vis2 = 0; continue; }      /* completed by a trivial column before this one */ \
        const uint64_t bit = 1ull << il;                                                                                           \
        VIS &= ~bit; ACC |= bit; k++;                                                                                              \
        const uint32_t c0 = __builtin_amdgcn_readlane(XA, il), c1 = __builtin_amdgcn_readlane(XB, il), c2 = __builtin_amdgcn_readlane(XC, il); \
        uint32_t a0 = c0 & ~lockU0, a1 = c1 & ~lockU1, a2 = c2 & ~lockU2;                                                          \
        if (__builtin_expect(!(a0 | a1 | a2), 0)) {            /* rare: nothing outside the locked and reserved rows -- steal, or dependent */ \
            const int p = (BASE) + il;                                                                                             \
            /* rows of columns that are still trivial and come after this position (lane = row; a robbed column's row is a pivot's) */ \
            const int pa = posrow[lane], pb = posrow[64 + (lane < 27 ? lane : 0)];                                                 \
            const bool ta = pa > p && pa < OSD_TRIV && (((pa < 64 ? triv0 >> pa : triv1 >> (pa - 64)) & 1ull) != 0);               \
            const bool tb = lane < 27 && pb > p && pb < OSD_TRIV && (((pb < 64 ? triv0 >> pb : triv1 >> (pb - 64)) & 1ull) != 0);  \
            const uint64_t s01 = (((uint64_t)c1 << 32) | c0) & __ballot(ta);                                                       \
            a0 = (uint32_t)s01; a1 = (uint32_t)(s01 >> 32); a2 = c2 & (uint32_t)__ballot(tb);                                      \
            if (a0 | a1 | a2) {                                                                                                    \
                const int r = s01 ? __builtin_ctzll(s01) : 64 + __builtin_ctz(a2);                                                 \
                const int q = __builtin_amdgcn_readfirstlane(posrow[r]);   /* the robbed trivial column: visited like any other from now on */ \
                if (q < 64) { triv0 &= ~(1ull << q); vis0 |= 1ull << q; tc0 -= (lane > q) ? 1 : 0; tc1 -= 1; }                    \
                else { triv1 &= ~(1ull << (q - 64)); vis1 |= 1ull << (q - 64); tc1 -= (lane > q - 64) ? 1 : 0; }                   \
                ntriv--;                                                                                                           \
            } else { ACC &= ~bit; k--; }                       /* dependent on the accepted columns */                             \
        }                                                                                                                          \
        asm volatile("" : "+s"(a2));                           /* opaque: nothing about the branch above is threaded into the code below */ \
        /* the accept step: pivot = the lowest eligible row, clear the column's other 1s everywhere, lock the row */                \
        if (a0) {                                                                                                                  \
            const uint32_t b = a0 & (0u - a0), bp = __builtin_ctz(a0), m0 = c0 & ~b, m1 = c1, m2 = c2;                             \
            UPD0                                                                                                                   \
            lockU0 |= b;                                                                                                           \
        } else if (a1) {                                                                                                           \
            const uint32_t b = a1 & (0u - a1), bp = __builtin_ctz(a1), m0 = c0, m1 = c1 & ~b, m2 = c2;                             \
            UPD1                                                                                                                   \
            lockU1 |= b;                                                                                                           \
        } else {                                               /* (a2 = 0: a dependent column -- bit 31 of the third word is never set: mask 0, no row locked) */ \
            const uint32_t b = a2 & (0u - a2), bp = __builtin_ctz(a2 | 0x80000000u), m0 = c0, m1 = c1, m2 = c2 & ~b;               \
            UPD2                                                                                                                   \
            lockU2 |= b;                                                                                                           \
        }                                                                                                                          \
    }
    uint32_t lockU0 = (uint32_t)lockU01, lockU1 = (uint32_t)(lockU01 >> 32);
    OT(2);
    OSD_STEP(x00, x01, x02, acc0, vis0, tc0, 0,
             OSD_UPD(x00, x00, x01, x02) OSD_UPD(x10, x10, x11, x12) OSD_UPD(x20, x20, x21, x22),
             OSD_UPD(x01, x00, x01, x02) OSD_UPD(x11, x10, x11, x12) OSD_UPD(x21, x20, x21, x22),
             OSD_UPD(x02, x00, x01, x02) OSD_UPD(x12, x10, x11, x12) OSD_UPD(x22, x20, x21, x22))
    OSD_STEP(x10, x11, x12, acc1, vis1, tc1, 64,
             OSD_UPD(x10, x10, x11, x12) OSD_UPD(x20, x20, x21, x22),
             OSD_UPD(x11, x10, x11, x12) OSD_UPD(x21, x20, x21, x22),
             OSD_UPD(x12, x10, x11, x12) OSD_UPD(x22, x20, x21, x22))
    {
        int tc2 = ntriv;                                      // every trivial position lies before register set 2
        OSD_STEP(x20, x21, x22, acc2, vis2, tc2, 128, OSD_UPD(x20, x20, x21, x22), OSD_UPD(x21, x20, x21, x22), OSD_UPD(x22, x20, x21, x22))
    }
#undef OSD_STEP
#undef OSD_UPD
    // the trivial columns of the basis: the first 91 - k of them
    acc0 |= __ballot(((triv0 >> lane) & 1ull) && tc0 < 91 - k);
    acc1 |= __ballot(((triv1 >> lane) & 1ull) && tc1 < 91 - k);
    __syncthreads();                                          // posrow (in skey) is done with: the flip rows overlay it below
#else
#define OSD_STEP(XA, XB, XC, ACC, IL)                                                                                              \
    {                                                                                                                              \
        OT_VISIT;                                                                                                                  \
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
#endif
    OT(3);
    // Every accepted column is now a unit vector (its pivot row).  Acceptance order = position order, so the accepted column at
    // position p is the kk-th accepted one with kk = number of accepted positions before p.
    const bool in0 = (acc0 >> lane) & 1ull, in1 = (acc1 >> lane) & 1ull, in2 = (acc2 >> lane) & 1ull;
    const int n0 = __popcll(acc0), n1 = __popcll(acc1);
    const int kk0 = __builtin_amdgcn_mbcnt_hi((uint32_t)(acc0 >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)acc0, 0));
    const int kk1 = n0 + __builtin_amdgcn_mbcnt_hi((uint32_t)(acc1 >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)acc1, 0));
    const int kk2 = n0 + n1 + __builtin_amdgcn_mbcnt_hi((uint32_t)(acc2 >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)acc2, 0));
    // flip i = the row locked by accepted column 90 - i (least reliable basis members first): that column's lane publishes the INDEX of its
    // pivot row (its column is the unit vector of that row); all 91 basis members exist, so every flip i < nflip <= 91 gets its entry
    int* fri = reinterpret_cast<int*>(frow);                   // [OSD_MAXFLIP] pivot row of flip i
#define OSD_ROWIDX(X0, X1, X2) ((X0) ? __builtin_ctz(X0) : (X1) ? 32 + __builtin_ctz(X1) : 64 + __builtin_ctz((X2) | 0x80000000u))
    { const int i = 90 - kk0; if (in0 && i >= 0 && i < nflip) fri[i] = OSD_ROWIDX(x00, x01, x02); }
    { const int i = 90 - kk1; if (in1 && i >= 0 && i < nflip) fri[i] = OSD_ROWIDX(x10, x11, x12); }
    { const int i = 90 - kk2; if (in2 && i >= 0 && i < nflip) fri[i] = OSD_ROWIDX(x20, x21, x22); }
#undef OSD_ROWIDX
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
    // Flip i's row is one bit position of the 91-bit column: its index is read as a SCALAR (a broadcast LDS read), the word it lies in
    // is a uniform three-way branch, and a column's flip bit costs a v_bfe_u32 and a v_lshl_or_b32 -- 6 vector instructions per flip
    // for the three register sets (the generic "column AND unit vector != 0" form: ~30; this loop was 900 of the ~5800 vector
    // instructions of an attempt, and the kernel is bound by their number: profiles/r05_notes.md).
    uint32_t f0l = 0, f1l = 0, f2l = 0;
    uint32_t f0h = (uint32_t)((__popc(x00 & hm0) + __popc(x01 & hm1) + __popc(x02 & hm2)) & 1) << 31,
             f1h = (uint32_t)((__popc(x10 & hm0) + __popc(x11 & hm1) + __popc(x12 & hm2)) & 1) << 31,
             f2h = (uint32_t)((__popc(x20 & hm0) + __popc(x21 & hm1) + __popc(x22 & hm2)) & 1) << 31;
#define OSD_FBIT(XA, XB, XC, FA, FB, FC, SH) { FA |= __builtin_amdgcn_ubfe(XA, bp, 1u) << (SH); FB |= __builtin_amdgcn_ubfe(XB, bp, 1u) << (SH); \
                                               FC |= __builtin_amdgcn_ubfe(XC, bp, 1u) << (SH); }
#define OSD_FLIPS(LO, HI, FA, FB, FC, SUB)                                                                                         \
    for (int i = (LO); i < (HI); i++) {                                                                                            \
        const int r = __builtin_amdgcn_readfirstlane(fri[i]);                                                                      \
        const uint32_t bp = (uint32_t)r & 31u, sh = (uint32_t)(i - (SUB));                                                         \
        if (r < 32) OSD_FBIT(x00, x10, x20, FA, FB, FC, sh)                                                                        \
        else if (r < 64) OSD_FBIT(x01, x11, x21, FA, FB, FC, sh)                                                                   \
        else OSD_FBIT(x02, x12, x22, FA, FB, FC, sh)                                                                               \
    }
    OSD_FLIPS(0, nflipA < 32 ? nflipA : 32, f0l, f1l, f2l, 0)
    OSD_FLIPS(32, nflipA, f0h, f1h, f2h, 32)
    // back to natural column order: ftab[column] (bit i = flip i has a 1 in this column, i < 62; bit 63 = the order-0 codeword bit)
    ftab[ord0] = ((uint64_t)f0h << 32) | f0l; ftab[ord1] = ((uint64_t)f1h << 32) | f1l; if (has2) ftab[ord2] = ((uint64_t)f2h << 32) | f2l;
    if (wide) {                                            // flips 62 .. nflip - 1 into the second word
        uint32_t g0 = 0, g1 = 0, g2 = 0;
        OSD_FLIPS(OSD_FLIPS_A, nflip, g0, g1, g2, OSD_FLIPS_A)
        ftabB[ord0] = g0; ftabB[ord1] = g1; if (has2) ftabB[ord2] = g2;
    }
#undef OSD_FLIPS
#undef OSD_FBIT
    __syncthreads();
    OT(5);
    // CRC syndromes (the CRC is linear): lane i < min(nflip, 62) takes flip i, lane 63 the order-0 codeword (flips 62.. in a second round).  The lane gathers its word as a
    // 91-bit column set (bit `bitsel` of every ftab entry), then each of the 14 syndrome bits is a masked parity (d_SYNM)
    {
        const int bitsel = (lane < nflipA) ? lane : 63;
        const uint32_t sh = (uint32_t)bitsel & 31u;
        // the lane reads only the 32-bit half of each entry that holds its bit (two addresses per read: both broadcast), then a
        // v_bfe_u32 and a v_lshl_or_b32 per entry
        const uint32_t* fh = reinterpret_cast<const uint32_t*>(ftab) + (bitsel >= 32 ? 1 : 0);
        uint32_t ra = 0, rb = 0, rc = 0;
#pragma unroll 8
        for (int v = 0; v < 32; v++) {
            ra |= __builtin_amdgcn_ubfe(fh[2 * v], sh, 1u) << v;
            rb |= __builtin_amdgcn_ubfe(fh[2 * (32 + v)], sh, 1u) << v;
            rc |= __builtin_amdgcn_ubfe(fh[2 * (64 + (v < 27 ? v : 0))], sh, 1u) << v;
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
// dead sort keys (5.6 KB allowed 7): 0.739 -> 0.710 ms per 256 frames (profiles/archive/r03_notes.md; the attribute alone, LDS-bound at 7: 0.781)
#ifndef OSD_ATTR
#define OSD_ATTR __attribute__((amdgpu_waves_per_eu(8, 8)))
#endif
#define OSD_KERNEL(NAME, WIDE)                                                                                                       \
__global__ __launch_bounds__(64) OSD_ATTR void NAME(int mode, const float* __restrict__ llr_in, const float* __restrict__ saved,             \
                                           const Att* __restrict__ attB, ft8rx_record* __restrict__ rec,                            \
                                           const int32_t* __restrict__ ncand, Att* __restrict__ attO,                               \
                                           ft8rx_event* ev, int32_t* evcount, const uint32_t* __restrict__ trials, int ntr,        \
                                           int nflip, int max_hd, WorkList work, WorkList nanlist) {                                \
    if (mode == 2) { osd_attempt<WIDE, false>(threadIdx.x, 2, blockIdx.x, llr_in, saved, attB, rec, ncand, attO, ev, evcount, trials, ntr, nflip, max_hd, nanlist); return; } \
    const int n = *work.count * 10;                                                                                                 \
    _Pragma("unroll 1")                                                                                                             \
    for (int item = blockIdx.x; item < n; item += gridDim.x) {                                                                      \
        int lane = threadIdx.x;                                                                                                     \
        asm volatile("" : "+v"(lane));       /* opaque per item: nothing lane-specific is hoisted across attempts (register pressure) */ \
        osd_attempt<WIDE, false>(lane, 0, work.items[item / 10] * 10 + item % 10, llr_in, saved, attB, rec, ncand, attO, ev, evcount, trials, ntr, nflip, max_hd, nanlist); \
        __syncthreads();                     /* the LDS arrays are reused by the next attempt */                                   \
    }                                                                                                                               \
}
OSD_KERNEL(k_osd, false)
OSD_KERNEL(k_osd_wide, true)        /* more than OSD_FLIPS_A flip rows */
#undef OSD_KERNEL
// the attempts the main kernels left on `nanlist` (attempt ids as they got them: candidate * 10 + slot, or the vector index in mode 2)
#define OSD_NAN_KERNEL(NAME, WIDE)                                                                                                   \
__global__ __launch_bounds__(64) void NAME(int mode, const float* __restrict__ llr_in, const float* __restrict__ saved,             \
                                           const Att* __restrict__ attB, ft8rx_record* __restrict__ rec,                            \
                                           const int32_t* __restrict__ ncand, Att* __restrict__ attO,                               \
                                           ft8rx_event* ev, int32_t* evcount, const uint32_t* __restrict__ trials, int ntr,        \
                                           int nflip, int max_hd, WorkList nanlist) {                                               \
    const int n = *nanlist.count;                                                                                                   \
    _Pragma("unroll 1")                                                                                                             \
    for (int item = blockIdx.x; item < n; item += gridDim.x) {                                                                      \
        osd_attempt<WIDE, true>(threadIdx.x, mode, nanlist.items[item], llr_in, saved, attB, rec, ncand, attO, ev, evcount, trials, ntr, nflip, max_hd, nanlist); \
        __syncthreads();                                                                                                            \
    }                                                                                                                               \
}
OSD_NAN_KERNEL(k_osd_nan, false)
OSD_NAN_KERNEL(k_osd_nan_wide, true)
#undef OSD_NAN_KERNEL
#define OSD_NAN_GRID 512             /* blocks of the NaN kernels: they stride over a list that is almost always empty */

#endif
