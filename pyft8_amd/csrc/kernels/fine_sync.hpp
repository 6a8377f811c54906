// fine_sync.hpp -- fine time/frequency sync (receiver.py:140-206)
// Part of libft8rx.so.  The IFFT stages and the symbol DFT are shared device code (k_refine3 of the main unit uses them); the
// kernel itself, k_fine, is built in the second translation unit only (ft8rx_ilp.hip, ILP scheduling: FT8RX_ILP_UNIT).
#ifndef FT8RX_FINE_SYNC_HPP
#define FT8RX_FINE_SYNC_HPP

// ------------------------------------------------------------------------------------ fine time/frequency sync (receiver.py:140-206)
// One 128-thread block per candidate.  The 3200-point inverse FFT (conj o forward o conj) runs in place in
// one LDS buffer as three register-fused stages [8] | [4,4] | [5,5]; stage 1 reads the tapered spectrum
// slice straight from global memory and knows that only 1000 of the 3200 bins are non-zero.  The 1/3200
// scale and the output conjugation are applied where the series is consumed.
#ifndef FINE_NT
#define FINE_NT 128
#endif
#ifndef FINE_WV
#define FINE_WV 3          /* <= 168 VGPRs: with 29.1 KB of LDS five 2-wave blocks fit a CU, i.e. three waves on two of its SIMDs */
#endif
#define FINE_INV 0.0003125f

// Timing-only instrumentation (build with -DFINE_TIMING, tools/fine_timing.py; never defined in the product): wave 0 of every block
// accumulates the shader cycles between consecutive marks into an LDS table and flushes it to g_fine_t[] once at the end.
#ifdef FINE_TIMING
__device__ unsigned long long g_fine_t[32];
#define FT_DECL __shared__ unsigned long long ft_l[32]; if (tid < 32) ft_l[tid] = 0; __syncthreads(); unsigned long long ft_prev = __builtin_readcyclecounter();
#define FT_ARG , unsigned long long& ft_prev, unsigned long long* ft_l
#define FT_PASS , ft_prev, ft_l
#define FT(i) do { if (tid == 0) { unsigned long long ft_now = __builtin_readcyclecounter(); ft_l[i] += ft_now - ft_prev; ft_prev = __builtin_readcyclecounter(); } } while (0)
#define FT_FLUSH do { __syncthreads(); if (tid < 32) atomicAdd(&g_fine_t[tid], ft_l[tid]); } while (0)
#else
#define FT_DECL
#define FT_ARG
#define FT_PASS
#define FT(i) do { } while (0)
#define FT_FLUSH do { } while (0)
#endif

// Input of the inverse transform = conj(taper * spec) over the rolled 3200-bin slice (receiver.py:180-185); bins 850 .. 3049 are zero.
// `sl` is the candidate's spectrum window staged in LDS: sl[i] = spec[fb0 - 182 + i], i < 1064 (covers every ftweak); off = ftweak + 182:
//   bin k < 850         = sl[off + k],              tapered by taper[k - 750] for k >= 750
//   bin k = 3050 + j    = sl[off - 150 + j],        tapered by taper[j] for j < 100
// Taper products are formed in fp64 and rounded once (the reference multiplies complex64 by a float64 ramp).
#define FINE_SLICE 1064
FT8_DEV cpx fine_taper(cpx v, double t) { return make_float2((float)((double)v.x * t), (float)((double)v.y * t)); }
FT8_DEV cpx fine_conj(cpx v) { return make_float2(v.x, -v.y); }

// forward FFT of the conjugated slice into z (unscaled, unconjugated), natural Stockham layout, in place.
// (Measured alternatives, profiles/archive/r01_notes.md: 256-thread blocks 7.96 ms, bank-conflict-free padded/transposed
// inter-stage layouts 6.92 ms, this version 6.36 ms per 256 frames: the kernel is latency/barrier bound.)
//
// Pass [8]: n = 3200, s = 1, m = 400: butterfly p reads bins p + 400 j, of which only j = 0, 1, (2 if p < 50), (7 if p >= 250) are
// non-zero.  Thread tid owns p = tid + 128 i, i < 4 (i = 3: tid < 16).  The code is STRAIGHT-LINE: every load of the stage is issued
// up front from clamped addresses and absent / untapered inputs are resolved by selects, never by divergent branches -- a branch
// around a dependent LDS or table load costs a full round trip per basic block (the r01 kernel had 13 of them in this stage; r02_notes).
// Which inputs exist / are tapered is static per round:
//   i = 0 (p <  128): bins p, p+400;            p+800 (tapered, index p + 50) for p < 50
//   i = 1 (p <  256): bins p, p+400;            p+2800 = 3050 + (p-250) (tapered) for p >= 250
//   i = 2 (p <  384): bins p, p+400 (tapered for p >= 350), p+2800 (tapered for p < 350): exactly one of the two is tapered
//   i = 3 (p <  400): bins p, p+400 (tapered, index p - 350), p+2800 (untapered)
// Twiddle multiplies are unconditional: W^0 = (1, -0) is an exact identity (the skip rule "j p = 0" of the contract only avoids it).
// dft<8> (ft8_dev.h) for the two input patterns of this stage, with the additions of exact zeros left out: the non-zero terms go
// through the same operations in the same order as in the generic butterfly, so every non-zero result is bit-identical (only the
// sign of an exact zero can differ: x + 0 turns -0 into +0, leaving the term out keeps -0 -- no output can see that).
//   inputs 0, 1, 2 (round 0):      56 -> 28 flops        inputs 0, 1, 7 (rounds 1..3):      56 -> 32 flops
FT8_DEV void dft8_in012(cpx* a) {
    const float h = 0.70710678f;
    const cpx a0 = a[0], a1 = a[1], a2 = a[2];
    const cpx e0 = cadd(a0, a2), e2 = csub(a0, a2);
    const cpx e1 = make_float2(a0.x + a2.y, a0.y - a2.x), e3 = make_float2(a0.x - a2.y, a0.y + a2.x);
    const cpx o1 = make_float2(h * (a1.x + a1.y), h * (a1.y - a1.x));
    const cpx o2 = make_float2(a1.y, -a1.x);
    const cpx o3 = make_float2(o1.y, -o1.x);                     // = (h (a1.y - a1.x), -(h (a1.x + a1.y)))
    a[0] = cadd(e0, a1); a[4] = csub(e0, a1);
    a[1] = cadd(e1, o1); a[5] = csub(e1, o1);
    a[2] = cadd(e2, o2); a[6] = csub(e2, o2);
    a[3] = cadd(e3, o3); a[7] = csub(e3, o3);
}
FT8_DEV void dft8_in017(cpx* a) {
    const float h = 0.70710678f;
    const cpx a0 = a[0], a1 = a[1], a7 = a[7];
    const cpx q0 = cadd(a1, a7), q2 = csub(a1, a7);
    const cpx q1 = make_float2(a1.x - a7.y, a1.y + a7.x), q3 = make_float2(a1.x + a7.y, a1.y - a7.x);
    const cpx o1 = make_float2(h * (q1.x + q1.y), h * (q1.y - q1.x));
    const cpx o2 = make_float2(q2.y, -q2.x);
    const cpx o3 = make_float2(h * (q3.y - q3.x), -(h * (q3.x + q3.y)));
    a[0] = cadd(a0, q0); a[4] = csub(a0, q0);
    a[1] = cadd(a0, o1); a[5] = csub(a0, o1);
    a[2] = cadd(a0, o2); a[6] = csub(a0, o2);
    a[3] = cadd(a0, o3); a[7] = csub(a0, o3);
}
static_assert(FINE_NT == 128, "fine_stage1 is written for 128 threads");
FT8_DEV void fine_stage1(const cpx* S, int fb, cpx* z, const cpx* __restrict__ W,
                         const double* __restrict__ taper, int tid FT_ARG) {
    const cpx zero = make_float2(0.0f, 0.0f);
    const cpx* s0 = S + fb + tid;                    // bin p of round i at s0[128 i]
    const cpx* s7 = S + fb - 150 - 250 + tid;        // bin p + 2800 = 3050 + (p - 250) of round i at s7[128 i]
    // ---- loads (LDS slice + taper table), all issued before the first butterfly
    const cpx a00 = s0[0], a01 = s0[400];
    const cpx a10 = s0[128], a11 = s0[528];
    const cpx a20 = s0[256], a21 = s0[656];
    const bool r3 = tid < 16;
    const int t3 = r3 ? tid : 0;
    const cpx a30 = s0[384 + t3 - tid], a31 = s0[784 + t3 - tid];
    const bool has02 = tid < 50;
    const cpx a02r = s0[has02 ? 800 : 0];
    const double t02 = taper[has02 ? tid + 50 : 0];
    const bool has17 = tid >= 122;                    // p = tid + 128 >= 250
    const cpx a17r = s7[has17 ? 128 : 250];           // (250: any in-range address, p = tid + 250)
    const double t17 = taper[has17 ? tid - 122 : 0];
    const uint32_t m02 = has02 ? 0xFFFFFFFFu : 0u, m17 = has17 ? 0xFFFFFFFFu : 0u;     // absent inputs: zeroed by masks, not by selects (r02_notes)
    const cpx a27r = s7[256];                          // p = tid + 256 >= 250 always
    const bool tap21 = tid >= 94;                      // p >= 350: bin p + 400 is tapered (index p - 350), else bin p + 2800 (index p - 250)
    const double t2 = taper[tap21 ? tid - 94 : tid + 6];
    const cpx a37r = s7[384 + t3 - tid];
    const double t31 = taper[34 + t3];                 // p + 400 - 750 = 34 + tid
    // ---- resolve
    const cpx t02v = fine_taper(a02r, t02), t17v = fine_taper(a17r, t17);
    const cpx a02 = make_float2(__uint_as_float(__float_as_uint(t02v.x) & m02), __uint_as_float(__float_as_uint(t02v.y) & m02));
    const cpx a17 = make_float2(__uint_as_float(__float_as_uint(t17v.x) & m17), __uint_as_float(__float_as_uint(t17v.y) & m17));
    const cpx sel2 = fine_taper(tap21 ? a21 : a27r, t2);
    const cpx b21 = tap21 ? sel2 : a21, a27 = tap21 ? a27r : sel2;
    const cpx b31 = fine_taper(a31, t31);
    cpx a[4][8];
#pragma unroll
    for (int i = 0; i < 4; i++)
#pragma unroll
        for (int j = 2; j < 7; j++) a[i][j] = zero;
    a[0][0] = fine_conj(a00); a[0][1] = fine_conj(a01); a[0][2] = fine_conj(a02); a[0][7] = zero;
    a[1][0] = fine_conj(a10); a[1][1] = fine_conj(a11); a[1][7] = fine_conj(a17);
    a[2][0] = fine_conj(a20); a[2][1] = fine_conj(b21); a[2][7] = fine_conj(a27);
    a[3][0] = fine_conj(a30); a[3][1] = fine_conj(b31); a[3][7] = fine_conj(a37r);
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const int p = tid + 128 * i;
        const int pw = (i < 3) ? p : 384 + t3;          // clamped: twiddle addresses stay in range for the idle threads of round 3
        cpx w[8];
#pragma unroll
        for (int j = 1; j < 8; j++) w[j] = W[j * pw];
        if (i == 0) dft8_in012(a[i]); else dft8_in017(a[i]);     // which inputs exist is static per round (above)
#pragma unroll
        for (int j = 1; j < 8; j++) a[i][j] = cmul(a[i][j], w[j]);
        if (i < 3 || r3) {
#pragma unroll
            for (int j = 0; j < 8; j++) z[8 * p + j] = a[i][j];
        }
    }
    FT(1);
    __syncthreads();
    FT(2);
}
FT8_DEV void fine_stage2(cpx* z, const cpx* w400, int tid FT_ARG) {
    typedef Fused2<3200, 400, 8, 4, 4> F;                         // passes [4,4]: n = 400, s = 8; 200 groups
    constexpr int R = (F::groups + FINE_NT - 1) / FINE_NT;
    cpx a[R][4][4];
    // group g = (pp = g / 8, q = g % 8): in  q + 8(pp + 25 j' + 100 j),  out  q + 8 j + 32 (4 pp + j')
#pragma unroll
    for (int r = 0; r < R; r++) { const int g = tid + FINE_NT * r; if (g < F::groups) F::load_affine<200, 800>(z, g, a[r]); }
    FT(3);
    __syncthreads();
    FT(4);
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
                const int pq = pp + 25 * jp;            // (pq = 0: W^0 = (1, -0), an exact identity -- no branch)
#pragma unroll
                for (int j = 1; j < 4; j++) a[r][jp][j] = cmul(a[r][jp][j], w400[j * pq]);
            }
#pragma unroll
            for (int j = 0; j < 4; j++) {
                cpx b[4];
#pragma unroll
                for (int jp = 0; jp < 4; jp++) b[jp] = a[r][jp][j];
                dft<4>(b);
#pragma unroll
                for (int jp = 1; jp < 4; jp++) b[jp] = cmul(b[jp], w400[4 * jp * pp]);
#pragma unroll
                for (int jp = 0; jp < 4; jp++) a[r][jp][j] = b[jp];
            }
            F::store_affine<32, 8>(z, (g & 7) + 128 * (g >> 3), a[r]);
        }
    }
    FT(5);
    __syncthreads();
    FT(6);
}
// Only output samples in [lo, hi) are needed (the scoring IFFTs read one Costas block = ~230 samples): a final
// radix-5 butterfly (q, j) produces samples q + 128 j + 640 j', at most one of which can fall in a window
// shorter than 640, so butterflies with no sample in the window are skipped.  Needed outputs are bit-identical.
FT8_DEV void fine_stage3(cpx* z, const cpx* __restrict__ W, int tid, int lo, int hi FT_ARG) {
    typedef Fused2<3200, 25, 128, 5, 5> F;                        // passes [5,5]: n = 25, s = 128; 128 groups
    constexpr int R = F::groups / FINE_NT;
    cpx a[R][5][5];
    // group q: in  q + 128 (j' + 5 j),  out  q + 128 j + 640 j'
#pragma unroll
    for (int r = 0; r < R; r++) F::load_affine<128, 640>(z, tid + FINE_NT * r, a[r]);
    FT(7);
    __syncthreads();
    FT(8);
#pragma unroll
    for (int r = 0; r < R; r++) {
        const int q = tid + FINE_NT * r;
        F::compute_passA(0, a[r], W);
#pragma unroll
        for (int j = 0; j < 5; j++) {
            const int rr = q + 128 * j;                                                 // < 640
            const int first = rr + 640 * (lo / 640 + (rr < lo % 640 ? 1 : 0));          // smallest rr + 640 j' >= lo (lo is block-uniform)
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
    FT(9);
    __syncthreads();
    FT(10);
}
FT8_DEV void fine_fft(const cpx* S, int fb, cpx* z, const cpx* w400, const Tables& T, int tid, int lo, int hi FT_ARG) {
    FT(0);
    fine_stage1(S, fb, z, T.W3200, T.taper, tid FT_PASS);
    fine_stage2(z, w400, tid FT_PASS);
    fine_stage3(z, T.W3200, tid, lo, hi FT_PASS);
}

// |32-pt DFT| tones 0..7 of the symbol starting at sample i0, computed by the 4 lanes of a quad
template <int NT>
FT8_DEV void fine_sym_quad(const cpx* z, int i0, int n2, const cpx* wq, float* mag) {
    if (i0 < 0) i0 = 0;
    if (i0 > 3168) i0 = 3168;
    cpx x[8];
#pragma unroll
    for (int n1 = 0; n1 < 8; n1++) { cpx v = z[i0 + 4 * n1 + n2]; x[n1] = make_float2(v.x * FINE_INV, -(v.y * FINE_INV)); }
    sym32_quad<NT>(x, n2, wq, mag);
}

#ifdef FT8RX_ILP_UNIT
// ---- score of a NON-ZERO frequency tweak straight from the spectrum slice, without its time series (round 4) ----
// The reference forms z = ifft(S) and scores |fft(z[i0 : i0 + 32])[t]| on the 7 symbols of the middle Costas block (receiver.py:186-206).
// Substituting one transform into the other -- an exact identity, all 1000 non-zero bins included --
//   T[s][t] = 1/3200 sum_k X[k] e^{2 pi i k nb0 / 3200} D(k - 100 t) e^{2 pi i k s / 100},   D(m) = sum_{n<32} e^{2 pi i n m / 3200},
// k = -150 .. 849 the rolled slice's bins, nb0 = the block's first sample.  With m = r + 100 d (0 <= r < 100) the Dirichlet kernel is a phase
// times a REAL number, D(m) = e^{i pi 31 r / 3200} g^d K(m), g = e^{-i pi / 32}, K(m) = sin(pi r / 100) / sin(pi m / 3200); the last factor of T
// has period 100 in k and E[s][100 - r] = conj E[s][r].  So with k = r + 100 j:
//   b[k]    = X[k] Phi[k],  Phi[k] = e^{2 pi i k nb0 / 3200} e^{i pi (31 r - 100 j) / 3200}   (per lane: the 10 phases of its bins, in registers)
//   H[t][r] = sum_j b[k] K(k - 100 t)          (step 1: lane per residue r, 10 bins, 7 tones, a complex-by-REAL correlation; K loaded once per d = j - t)
//   T[s][t] = 1/3200 sum_{p=0..25} cos_ps (P_p + (-1)^s P_{50-p}) + i sin_ps (M_p - (-1)^s M_{50-p}),   P_x, M_x = H[t][x] +- H[t][100-x]
//                                              (step 2: 16 lanes per tone, 26 items of four residues, four real multiply-adds per item and
//                                               symbol, a DPP row sum)
// -- the factor g^{-t} left over has modulus 1 and only |T| is scored.  3.5 k + 2.4 k complex-multiply equivalents instead of a pruned
// 3200-point IFFT and seven symbol DFTs.  The arithmetic (operation order, named fmas, the reduction tree) is the contract of
// oracle/ft8_oracle.c: fine_fscore -- bit-exact.
// What a lane of step 1 needs besides the slice is the same for every tweak of a candidate and lives in registers: its 16 values of K
// (FsLane::k, loaded before the time scan is scored so that the latency hides), its 10 phases Phi (after the time tweak is known) and
// its taper value.  In the IFFT image (dead between the time scan and the final transform), in complex slots: (cos, sin) (6 x 26),
// H (2 x 7 x 100, double-buffered), 8 x 7 x 8 magnitudes.
#define FS_CS 0
#define FS_H 156
#define FS_MAG 1556
struct FsLane {
    float k[17];        // K(r + 100 d), d = jlo - 7 .. jlo + 9 (the first one only for tone 7 of the final grid)
    cpx g0[10];         // G[k_q], k_q = r + 100 (jlo + q)
    cpx g[10];          // Phi[k_q] = cmul(conj W3200[(k_q nb0) mod 3200], G[k_q]) for the current nb0
    double tap;         // taper of the lane's first and last bin (the same value: k_0 + 150 = k_9 - 750)
    cpx cs[2];          // this thread's part of the (cos, sin) table on its way to the image
    int r, jlo;
};
FT8_DEV void fscore_fetch(FsLane& L, const Tables& T, int tid) {            // request everything that does not depend on the time tweak
    const int lane = tid & 63;
    const int ln = lane < 50 ? lane : 0;                                     // (lanes 50 .. 63 of a wave idle in step 1)
    L.r = tid < 64 ? 50 + ln : ln;
    L.jlo = tid < 64 ? -2 : -1;
#pragma unroll
    for (int u = 0; u < 17; u++) L.k[u] = T.K32[900 + L.r + 100 * (L.jlo - 7 + u)];
#pragma unroll
    for (int q = 0; q < 10; q++) L.g0[q] = T.G1000[L.r + 100 * (L.jlo + q) + 150];
    L.tap = T.taper[tid < 64 ? ln : 50 + ln];
#pragma unroll
    for (int u = 0; u < 2; u++) { const int i = tid + FINE_NT * u; L.cs[u] = T.CS100[i < 156 ? i : 0]; }
}
FT8_DEV void fscore_phases(FsLane& L, const Tables& T, int nb0) {          // the lane's ten phases for a block starting at sample nb0 (any sign)
    cpx w[10];
    {   // W3200[(k_q nb0) mod 3200], k_q = r + 100 (jlo + q): one reduction for the lane's first bin, then steps of (100 nb0) mod 3200
        const int step = (((100 * nb0) % 3200) + 3200) % 3200;
        int idx = ((((L.r + 100 * L.jlo) * nb0) % 3200) + 3200) % 3200;
#pragma unroll
        for (int q = 0; q < 10; q++) { w[q] = T.W3200[idx]; idx += step; idx -= (idx >= 3200) ? 3200 : 0; }
    }
#pragma unroll
    for (int q = 0; q < 10; q++) L.g[q] = cmul(make_float2(w[q].x, -w[q].y), L.g0[q]);
}
FT8_DEV void fscore_prepare(FsLane& L, cpx* zi, const Tables& T, int nb0, int tid) {
    fscore_phases(L, T, nb0);
#pragma unroll
    for (int u = 0; u < 2; u++) { const int i = tid + FINE_NT * u; if (i < 156) zi[FS_CS + i] = L.cs[u]; }
    __syncthreads();
}
// step 1 for NTONE tones (7: a frequency tweak's score; 8: the final grid)
template <int NTONE>
FT8_DEV void fscore_p1(const cpx* S, int off, const FsLane& L, cpx* H, int tid FT_ARG) {
    cpx b[10];
#pragma unroll
    for (int q = 0; q < 10; q++) {
        cpx x = S[off + L.r + 100 * (L.jlo + q)];
        if (q == 0 || q == 9) x = fine_taper(x, L.tap);
        b[q] = cmul(x, L.g[q]);
    }
    FT(16);
    // H_t = sum_q b[q] K(k_q - 100 t), q ascending; walked by d = j - t = jlo - 7 + u so that every K value is used from its register:
    // q = u + t - 7, and for a given tone the terms still arrive in ascending q
    float hx[NTONE], hy[NTONE];
#pragma unroll
    for (int t = 0; t < NTONE; t++) { hx[t] = 0.0f; hy[t] = 0.0f; }
#pragma unroll
    for (int u = 0; u < 17; u++) {
#pragma unroll
        for (int t = 0; t < NTONE; t++) {
            const int q = u + t - 7;
            if (q >= 0 && q < 10) {
                hx[t] = __builtin_fmaf(b[q].x, L.k[u], hx[t]);
                hy[t] = __builtin_fmaf(b[q].y, L.k[u], hy[t]);
            }
        }
    }
    FT(17);
#pragma unroll
    for (int t = 0; t < NTONE; t++) H[t * 100 + L.r] = make_float2(hx[t], hy[t]);
}
// forward 10-point DFT by the prime-factor map (no twiddles between the radix-2 and the radix-5 part): oracle/ft8_oracle.c: dft10_fwd
FT8_DEV void dft10_fwd(const cpx* x, cpx* y) {
    cpx a0[5], a1[5];
#pragma unroll
    for (int n = 0; n < 5; n++) { const cpx u = x[(2 * n) % 10], v = x[(5 + 2 * n) % 10]; a0[n] = cadd(u, v); a1[n] = csub(u, v); }
    dft<5>(a0); dft<5>(a1);
#pragma unroll
    for (int k = 0; k < 5; k++) { y[(6 * k) % 10] = a0[k]; y[(5 + 6 * k) % 10] = a1[k]; }
}
FT8_DEV float row16_sum(float v) {                 // sum over the 16 lanes of a DPP row, the same value on every lane: ((p0+p1)+(p2+p3)) + ... as a binary tree
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xB1, 0xf, 0xf, false));      // quad_perm:[1,0,3,2]
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x4E, 0xf, 0xf, false));      // quad_perm:[2,3,0,1]
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x141, 0xf, 0xf, false));     // row_half_mirror (quads hold equal values)
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x140, 0xf, 0xf, false));     // row_mirror (halves hold equal values)
    return v;
}
// The 49 magnitudes of tweak number n (n = 0 .. 7 in scan order) -> mags[n][s * 8 + t].  H is double-buffered by the parity of n, so ONE
// barrier per tweak is enough: a wave writes H[n & 1] for tweak n + 2 only after it has passed the barrier of tweak n + 1, which the
// other wave reaches after it has read H[n & 1] for tweak n.  The scores are formed after the scan, for all tweaks at once.
FT8_DEV void fine_fscore(const cpx* S, int off, const FsLane& L, cpx* zi, int n, int tid FT_ARG) {
    cpx* H = zi + FS_H + (n & 1) * 700;
    float* mags = reinterpret_cast<float*>(zi + FS_MAG) + n * 56;
    const int lane = tid & 63;
    if (lane < 50) fscore_p1<7>(S, off, L, H, tid FT_PASS);
    FT(11);
    __syncthreads();
    FT(12);
    {
        const int c = tid & 15, t = tid >> 4;
        cpx acc[7];
#pragma unroll
        for (int s = 0; s < 7; s++) acc[s] = make_float2(0.0f, 0.0f);
        const cpx* Ht = H + (t < 7 ? t : 0) * 100;
#pragma unroll
        for (int i = 0; i < 2; i++) {
            const int p = c + 16 * i;                          // item: the residues p, 100 - p, 50 - p, 50 + p (p = 0: 0 and 50; p = 25: 25 and 75)
            if (i < 1 || c < 10) {
                const bool first = (i == 0 && c == 0), last = (i == 1 && c == 9);
                const cpx zero = make_float2(0.0f, 0.0f);
                const cpx h1 = Ht[p];
                cpx h2 = Ht[first ? 99 : 100 - p], h3 = Ht[50 - p], h4 = Ht[50 + p];
                if (first) { h2 = zero; h4 = zero; }
                if (last) { h3 = zero; h4 = zero; }
                const cpx P = make_float2(h1.x + h2.x, h1.y + h2.y), M = make_float2(h1.x - h2.x, h1.y - h2.y);
                const cpx P2 = make_float2(h3.x + h4.x, h3.y + h4.y), M2 = make_float2(h3.x - h4.x, h3.y - h4.y);
                const cpx A = make_float2(P.x + P2.x, P.y + P2.y), B = make_float2(P.x - P2.x, P.y - P2.y);
                const cpx C = make_float2(M.x - M2.x, M.y - M2.y), D = make_float2(M.x + M2.x, M.y + M2.y);
                acc[0] = cadd(acc[0], A);
#pragma unroll
                for (int s = 1; s < 7; s++) {
                    const cpx e = zi[FS_CS + (s - 1) * 26 + p];
                    const cpx X = (s & 1) ? B : A, Y = (s & 1) ? D : C;
                    acc[s].x = __builtin_fmaf(X.x, e.x, __builtin_fmaf(-Y.y, e.y, acc[s].x));
                    acc[s].y = __builtin_fmaf(X.y, e.x, __builtin_fmaf(Y.x, e.y, acc[s].y));
                }
            }
        }
        FT(19);
#pragma unroll
        for (int s = 0; s < 7; s++) { acc[s].x = row16_sum(acc[s].x); acc[s].y = row16_sum(acc[s].y); }
        {   // every lane forms a magnitude and stores it -- lanes without one (c >= 7; tone 7) into the unused tone-7 slots: with the
            // store under `if (t < 7 && c < 7)` the compiler sinks the last row-sum step into the branch, where the DPP operand can no
            // longer be fused into the add (14 v_mov_dpp + 14 zero moves per tweak)
            cpx v = acc[0];
#pragma unroll
            for (int s = 1; s < 7; s++) v = (c == s) ? acc[s] : v;
            const float re = v.x * FINE_INV, im = v.y * FINE_INV;
            mags[c < 7 ? c * 8 + t : 7] = sqrtf(re * re + im * im);
        }
    }
    FT(14);
}

FT8_DEV void fine_candidate(int tid, int bid, const cpx* __restrict__ spec, ft8rx_record* __restrict__ rec,
                            const int32_t* __restrict__ ncand, float* __restrict__ llr0, const Tables& T, const ft8rx_config& cfg,
                            const int32_t* __restrict__ trip, int32_t* __restrict__ t_out /*[n][5]*/,
                            float* __restrict__ t_sd, float* __restrict__ t_sgrid) {
    __shared__ cpx z[3200];
    // The candidate's 1064 spectrum bins (the slice every tweak reads) are NOT staged in LDS beside the transform image: the time scan's
    // first stage reads its 13 values per thread straight from global memory, and while the time scan is scored the slice is copied
    // global -> LDS by the DMA path (global_load_lds_dwordx4: no registers, no wait) into the part of the image the scoring does not
    // read -- below or above it, by the candidate's start time -- for the frequency scan and the final grid, whose tables take the
    // other part.  37.6 -> 29.1 KB of LDS per block = 5 blocks (10 waves) per CU instead of 4 (profiles/r05_notes.md section 6).
    __shared__ cpx w400[400];          // W3200[8 t]: every twiddle of the [4,4] stage of the ONE transform (the time scan's); dead after it, then
    float* mg = reinterpret_cast<float*>(w400);      // [640] scoring (on, off) sums as fp64, later the [79][8] grid -- first written after that transform
    static_assert(sizeof(cpx) * 400 >= sizeof(float) * 640, "mg overlays w400");
    // (keeping w400 apart from mg and staging it once per block would save four loads and a barrier per candidate and still fit five
    // blocks per CU, but hipcc's register allocator crashes on that form under the ILP scheduling strategy: ROCm 7.2)
    __shared__ cpx w32[32];
    __shared__ float sc[16];
    __shared__ int ish[4];
    const int lane = tid & 63;
    int frame, ci = 0, f0, h0;
    if (trip) { frame = trip[3 * bid]; f0 = trip[3 * bid + 1]; h0 = trip[3 * bid + 2]; }
    else {
        frame = bid / MAXC; ci = bid % MAXC;
        if (ci >= ncand[frame]) return;
        const ft8rx_record& r = rec[(size_t)frame * MAXC + ci];
        if (r.status != FT8RX_ST_ACTIVE) return;
        f0 = r.f0_idx; h0 = r.h0_idx;
    }
    if (h0 < FT8RX_MIN_H0_FD || h0 > FT8RX_MAX_H0_FD) return;        // the middle Costas block leaves the series: k_fine_td's candidate
    const cpx w32v = T.W32[tid & 31];                             // requested with the twiddle loads below, stored before the barrier
    const int fb0 = 50 * f0;                                      // int(0.5 + fHz*16)
    {
        // the stage-2 twiddles (and the symbol-DFT twiddles) go to LDS; the slice stays in global memory for the time scan
        constexpr int NW = (400 + FINE_NT - 1) / FINE_NT;
        cpx wv[NW];
#pragma unroll
        for (int q = 0; q < NW; q++) { const int i = tid + FINE_NT * q; wv[q] = T.W3200[8 * (i < 400 ? i : 0)]; }
#pragma unroll
        for (int q = 0; q < NW; q++) { const int i = tid + FINE_NT * q; if (i < 400) w400[i] = wv[q]; }
        if (tid < 32) w32[tid] = w32v;
        __syncthreads();
    }
    const cpx* __restrict__ Sg = spec + (size_t)frame * FT8RX_SPEC_BINS + (fb0 - 182);      // global: slice bin i at Sg[i]
    FT_DECL
    cpx wq[8];
    sym32_twiddles(w32, tid & 3, wq);                             // every symbol DFT of this thread uses n2 = tid & 3
    const int tb0 = 8 * h0 + (h0 < 0 ? 1 : 0);                    // int(0.5 + tsec/0.005) truncates toward zero
    // Score of one Costas block (contract): per symbol a the quad leader forms on_a = |tone costas[a]| and off_a = sum of
    // the other six tones (b ascending) in fp64 from its registers; after ONE barrier every thread combines
    // S1 = sum_a on_a, S2 = sum_a off_a (a ascending) and score = (float)(S1 + w6 S2) -- no serial chain, no broadcast.
    double* dsum = reinterpret_cast<double*>(mg);              // [8][7][2] (on, off); mg is free until the final grid
    // --- time tweaks at ftweak 0: range(-8,8,2) -> 8 x 7 symbols, 4 lanes each
    fine_fft(Sg, 182, z, w400, T, tid, tb0 - 8 + 32 * 36, tb0 + 6 + 32 * 43 FT_PASS);   // the 8 time tweaks of the middle Costas block
    FsLane L;
    fscore_fetch(L, T, tid);                                   // the frequency scan's constants: requested here, used after the time scan
    // The scoring below reads samples [tb0 + 1144, tb0 + 1382) of the image (clamped).  The slice copy -- 9 chunks of 64 lanes x 16 bytes,
    // the last one running 88 bins past the slice on both sides -- goes to z[0, 1152) when that range lies above it, else to
    // z[2048, 3200); the frequency scan's tables (1780 slots) and the final grid's (1600) start at z[1152) resp. z[0).
    const bool lowslice = tb0 >= 667;                          // block-uniform
    cpx* slice = z + (lowslice ? 0 : 2048);
    cpx* zi = z + (lowslice ? 1152 : 0);
    static_assert(9 * 128 <= 1152 && 2048 + 9 * 128 <= 3200 && 1152 + 1780 <= 3200 && 1780 <= 2048, "slice copy and scan tables share the image");
    {
        const int wv = tid >> 6, ln = tid & 63;
#pragma unroll
        for (int c = 0; c < 5; c++) {
            const int ch = wv + 2 * c;                         // chunk: bins [128 ch, 128 ch + 128)
            if (ch < 9)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(Sg + 128 * ch + 2 * ln),
                                                 (__attribute__((address_space(3))) void*)(slice + 128 * ch), 16, 0, 0);
        }
    }
    const cpx* S = slice;                                      // valid after the wait + barrier in front of the frequency scan
    float* p = reinterpret_cast<float*>(slice);                // [464] the slice is dead once the final grid exists: reuse it
    float* llr = p + 464;                                      // [176]
    float* sq = llr + 176;                                     // [176]
#pragma unroll 1
    for (int r = 0; r < (224 + FINE_NT - 1) / FINE_NT; r++) {
        const int task = tid + FINE_NT * r, qd = task >> 2, n2 = task & 3;
        const bool valid = qd < 56;
        const int ti = valid ? qd / 7 : 0, a = valid ? qd - 7 * ti : 0;
        float mag[8];
        fine_sym_quad<7>(z, tb0 - 8 + 2 * ti + 32 * (36 + a), n2, wq, mag);
        // the magnitudes are pinned in front of the branch: sunk into it (only the quad leaders use them), the quad sums lose their
        // fused DPP operands (42 v_mov_dpp per round)
#pragma unroll
        for (int b = 0; b < 7; b++) asm volatile("" :: "v"(mag[b]));
        if (valid && n2 == 0) {
            const int c = d_COSTAS[a];
            double off = 0.0, on = 0.0;         // branch-free: adding +0.0 for the Costas tone leaves the running sum unchanged
#pragma unroll
            for (int b = 0; b < 7; b++) { const double m = (double)mag[b]; on = (b == c) ? m : on; off += (b == c) ? 0.0 : m; }
            dsum[(ti * 7 + a) * 2] = on; dsum[(ti * 7 + a) * 2 + 1] = off;
        }
    }
    FT(25);
    __syncthreads();
    int tt = -8; float score_f0 = 0.0f;
    {   // lane ti (mod 8) of every wave sums the block at time tweak ti; the first maximum (np.argmax) is then picked from the 8 lanes
        const int ti = tid & 7;
        double s1 = 0.0, s2 = 0.0;
#pragma unroll
        for (int a = 0; a < 7; a++) { s1 += dsum[(ti * 7 + a) * 2]; s2 += dsum[(ti * 7 + a) * 2 + 1]; }
        const float sct = (float)(s1 + W6 * s2);
#pragma unroll
        for (int u = 0; u < 8; u++) {
            const float v = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(sct), u));
            if (u == 0 || v > score_f0) { score_f0 = v; tt = -8 + 2 * u; }
        }
    }
    FT(26);
    // --- frequency tweaks: range(-32,33,8)
    float best = 0.0f; int ft = 0;
    __builtin_amdgcn_s_waitcnt(0x0F70);                        // vmcnt(0): the slice copy has landed (made visible by fscore_prepare's barrier)
    fscore_prepare(L, zi, T, tb0 + tt + 32 * 36, tid);         // the time scan is done with the series: the image takes the tables of the frequency scan
    FT(18);
#pragma unroll 1
    for (int n = 0; n < 8; n++) {                              // tweak 0 is the time scan's winner: same series, same offset, identical value
        fine_fscore(S, 182 - 32 + 8 * (n < 4 ? n : n + 1), L, zi, n, tid FT_PASS);
        FT(15);
    }
    __syncthreads();
    if (tid < 56) {                                            // (on, off) of symbol a of tweak n: the contract of the time scan, b ascending in fp64
        const int n = tid / 7, a = tid - 7 * n, c = d_COSTAS[a];
        const float* mags = reinterpret_cast<const float*>(zi + FS_MAG) + n * 56 + a * 8;
        double off = 0.0, on = 0.0;
#pragma unroll
        for (int b = 0; b < 7; b++) { const double m = (double)mags[b]; on = (b == c) ? m : on; off += (b == c) ? 0.0 : m; }
        dsum[tid * 2] = on; dsum[tid * 2 + 1] = off;
    }
    __syncthreads();
    {   // lane n (mod 8) of every wave sums tweak n; the first maximum in scan order, the winner of the time scan at its place in the middle
        const int n = tid & 7;
        double s1 = 0.0, s2 = 0.0;
#pragma unroll
        for (int a = 0; a < 7; a++) { s1 += dsum[(n * 7 + a) * 2]; s2 += dsum[(n * 7 + a) * 2 + 1]; }
        const float sct = (float)(s1 + W6 * s2);
#pragma unroll
        for (int i = 0; i < 9; i++) {
            const float v = (i == 4) ? score_f0 : __int_as_float(__builtin_amdgcn_readlane(__float_as_int(sct), i < 4 ? i : i - 1));
            if (i == 0 || v > best) { best = v; ft = -32 + 8 * i; }
        }
    }
    FT(20);
    // --- the 79 x 8 grid of the chosen tweaks, straight from the slice as well (oracle/ft8_oracle.c: fine_grid_freq): H for eight tones with
    // nb0 = the first sample of symbol 0, then |sum_r H[t][r] e^{2 pi i r s / 100}| for s = 0 .. 78 as the forward 100-point DFT of conj H,
    // 100 = 10 x 10 (r = 10 r1 + r2, s = s1 + 10 s2): 80 lanes transform the ten r1 of their (tone, r2), twiddle, 80 lanes the ten r2 of their
    // (tone, s1).  No 3200-point transform, no symbol DFTs: 0.5 k instead of 1.4 k instructions per wave.
    const int tb = tb0 + tt;
    cpx* Hc = zi;                      // [8][100]
    cpx* Ab = zi + 800;                // [8][10 s1][10 r2]
    {
        const int t = (tid < 80) ? tid / 10 : 0, c10 = (tid < 80) ? tid - 10 * t : 0;
        cpx tw[9];
#pragma unroll
        for (int s1 = 1; s1 < 10; s1++) tw[s1 - 1] = T.TW100[c10 * s1];           // requested now, used after the first barrier
        fscore_phases(L, T, tb);
        if (lane < 50) fscore_p1<8>(S, 182 + ft, L, Hc, tid FT_PASS);
        __syncthreads();
        FT(21);
        if (tid < 80) {                                                           // (tone t, r2 = c10): over r1
            cpx x[10], y[10];
#pragma unroll
            for (int r1 = 0; r1 < 10; r1++) { const cpx h = Hc[t * 100 + 10 * r1 + c10]; x[r1] = make_float2(h.x, -h.y); }
            dft10_fwd(x, y);
            Ab[t * 100 + c10] = y[0];
#pragma unroll
            for (int s1 = 1; s1 < 10; s1++) Ab[t * 100 + 10 * s1 + c10] = cmul(y[s1], tw[s1 - 1]);
        }
        __syncthreads();
        FT(22);
        if (tid < 80) {                                                           // (tone t, s1 = c10): over r2
            cpx x[10], y[10];
#pragma unroll
            for (int r2 = 0; r2 < 10; r2++) x[r2] = Ab[t * 100 + 10 * c10 + r2];
            dft10_fwd(x, y);
#pragma unroll
            for (int s2 = 0; s2 < 8; s2++) {
                const int sy = c10 + 10 * s2;
                const float re = y[s2].x * FINE_INV, im = y[s2].y * FINE_INV;
                if (sy < 79) mg[sy * 8 + t] = sqrtf(re * re + im * im);
            }
        }
    }
    // the reference clamps a symbol's first sample to [0, 3168]: the symbols before sample 0 all read the samples 0 .. 31, symbol 78 beyond
    // 3168 the last 32 -- their rows are the row of that position (block-uniform; a candidate has at most one of the two)
    // A symbol that starts exactly AT a clamp position reads the same samples as the clamped ones (bit-identical rows in the reference:
    // exact |LLR| ties for osd_012's argsort), so it takes the clamp row whenever some symbol lies strictly beyond; alone it is ordinary.
    const int n_lo = (tb < 0) ? min(79, (-tb) / 32 + 1) : 0;        // symbols 0 .. n_lo - 1 start at or before sample 0 (tb < 0: symbol 0 strictly)
    const int s_strict = (tb > 3168) ? 0 : (3168 - tb) / 32 + 1;    // first symbol strictly beyond sample 3168
    const int s_up = (s_strict < 79) ? max(0, (3168 - tb + 31) / 32) : 79;  // symbols s_up .. 78 start at or beyond 3168 (default search range: at most symbol 78)
    if (n_lo > 0 || s_up < 79) {
        __syncthreads();                                             // the rows above are written, Hc is free
        fscore_phases(L, T, n_lo > 0 ? 0 : 3168);
        if (lane < 50) fscore_p1<8>(S, 182 + ft, L, Hc, tid FT_PASS);
        __syncthreads();
        const int c = tid & 15, t = tid >> 4;
        float ax = 0.0f, ay = 0.0f;
#pragma unroll
        for (int i = 0; i < 7; i++) {
            if (i < 6 || c < 4) { const cpx h = Hc[t * 100 + c + 16 * i]; ax = ax + h.x; ay = ay + h.y; }
        }
        ax = row16_sum(ax); ay = row16_sum(ay);
        const float re = ax * FINE_INV, im = ay * FINE_INV;
        const float m = sqrtf(re * re + im * im);
        if (n_lo > 0) { for (int sy = c; sy < n_lo; sy += 16) mg[sy * 8 + t] = m; }
        else { for (int sy = s_up + c; sy < 79; sy += 16) mg[sy * 8 + t] = m; }
    }
    __syncthreads();
    FT(23);
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
    FT(24);
    if (trip && t_sgrid) for (int i = tid; i < 632; i += FINE_NT) t_sgrid[(size_t)bid * 632 + i] = mg[i];
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
        float* out = llr0 + (size_t)bid * 174;
        for (int i = tid; i < 174; i += FINE_NT) out[i] = llr[i];
    }
    FT(13);
    FT_FLUSH;
    if (tid == 0) {
        if (trip) { int32_t* o = t_out + 5 * (size_t)bid; o[0] = ret; o[1] = tt; o[2] = ft; o[3] = nsync; o[4] = snr; t_sd[bid] = sd; }
        else {
            ft8rx_record& r = rec[(size_t)frame * MAXC + ci];
            r.ttweak = (int8_t)tt; r.ftweak = (int8_t)ft; r.nsync = (uint8_t)nsync;
            if (ret == 0) r.status = FT8RX_ST_STOP_COSTAS;
            else { r.fine_sd = sd; r.snr_fine = (int8_t)snr; if (ret < 0) r.status = FT8RX_ST_STOP_FINE_SD; }
        }
    }
}


// test entry (trip != nullptr): one block per (frame, f0, h0) triple.  Pipeline: blocks stride over the fine-sync work list.
__global__ __launch_bounds__(FINE_NT, FINE_WV) void k_fine(const cpx* __restrict__ spec, ft8rx_record* __restrict__ rec,
                                                  const int32_t* __restrict__ ncand, float* __restrict__ llr0, Tables T, ft8rx_config cfg,
                                                  const int32_t* __restrict__ trip, int32_t* __restrict__ t_out /*[n][5]*/,
                                                  float* __restrict__ t_sd, float* __restrict__ t_sgrid, WorkList work) {
    if (trip) { fine_candidate(threadIdx.x, blockIdx.x, spec, rec, ncand, llr0, T, cfg, trip, t_out, t_sd, t_sgrid); return; }
    const int n = *work.count;
    // XCD-aware order: consecutive workgroup ids go round-robin over the 8 XCDs, the list holds a frame's candidates next to each other,
    // and their spectrum slices overlap -- so XCD x takes runs of FINE_RUN consecutive items (run = 8 k + x) and a frame's slices are
    // fetched into one L2 instead of eight (the launch grid is a multiple of 8)
#ifndef FINE_RUN
#define FINE_RUN 256
#endif
    const int nu = (n + 8 * FINE_RUN - 1) / (8 * FINE_RUN) * (8 * FINE_RUN);
#pragma unroll 1
    for (int u = blockIdx.x; u < nu; u += gridDim.x) {
        const int x = u & 7, j = u >> 3;
        const int item = (j / FINE_RUN) * (8 * FINE_RUN) + x * FINE_RUN + (j % FINE_RUN);
        if (item >= n) continue;
        int tid = threadIdx.x;
        asm volatile("" : "+v"(tid));                               // opaque per item: nothing thread-specific is hoisted across candidates (register pressure)
        fine_candidate(tid, work.items[item], spec, rec, ncand, llr0, T, cfg, nullptr, nullptr, nullptr, nullptr);
        __syncthreads();                                            // the LDS images are reused by the next candidate
    }
}
#endif  // FT8RX_ILP_UNIT

#ifndef FT8RX_ILP_UNIT
// ------------------------------------------------------------------------------------ fine sync of FAR-OUT candidates, in the time domain
// The frequency-domain scores and grid of k_fine are identities of the reference's IFFT + symbol DFTs as long as the middle Costas
// block of every tweak lies inside the 3200-sample series: h0 in [FT8RX_MIN_H0_FD, FT8RX_MAX_H0_FD].  The reference takes any
// search_time_range (receiver.py:312, 319) and clamps every symbol read to [0, 3168] (:189-195); a candidate further out is scored here
// the way the reference scores it (contract: oracle/ft8_oracle.c ft8o_fine, `far_out`): ONE series per tweak -- the time scan's, then one
// per non-zero frequency tweak, then the chosen one in full for the 79 x 8 grid -- and every symbol's 32-sample DFT at its clamped
// position.  Ten transforms per candidate (round 3's kernel did that for every candidate); such candidates only exist with a
// search_time_range beyond -6.1 .. +8.3 s, the launch is skipped otherwise.  Main translation unit (default scheduler).
FT8_DEV int fine_clamp_pos(int i0) { return i0 < 0 ? 0 : (i0 > 3168 ? 3168 : i0); }
FT8_DEV void fine_td_candidate(int tid, int bid, const cpx* __restrict__ spec, ft8rx_record* __restrict__ rec,
                               const int32_t* __restrict__ ncand, float* __restrict__ llr0, const Tables& T, const ft8rx_config& cfg,
                               const int32_t* __restrict__ trip, int32_t* __restrict__ t_out, float* __restrict__ t_sd, float* __restrict__ t_sgrid) {
    __shared__ cpx z[3200];
    __shared__ cpx w400[400];
    __shared__ cpx w32[32];
    __shared__ float mg[640];
    __shared__ double dsum[8 * 7 * 2];
    __shared__ float p[464], llr[176], sq[176], sc[16];
    __shared__ int ish[4];
    int frame, ci = 0, f0, h0;
    if (trip) { frame = trip[3 * bid]; f0 = trip[3 * bid + 1]; h0 = trip[3 * bid + 2]; }
    else {
        frame = bid / MAXC; ci = bid % MAXC;
        if (ci >= ncand[frame]) return;
        const ft8rx_record& r = rec[(size_t)frame * MAXC + ci];
        if (r.status != FT8RX_ST_ACTIVE) return;
        f0 = r.f0_idx; h0 = r.h0_idx;
        if (h0 >= FT8RX_MIN_H0_FD && h0 <= FT8RX_MAX_H0_FD) return;  // k_fine's candidate
    }
    for (int i = tid; i < 400; i += FINE_NT) w400[i] = T.W3200[8 * i];
    if (tid < 32) w32[tid] = T.W32[tid];
    __syncthreads();
    const int fb0 = 50 * f0;
    const cpx* __restrict__ Sg = spec + (size_t)frame * FT8RX_SPEC_BINS + (fb0 - 182);
    cpx wq[8];
    sym32_twiddles(w32, tid & 3, wq);
    const int tb0 = 8 * h0 + (h0 < 0 ? 1 : 0);
    // --- time tweaks at ftweak 0 (8 x 7 symbols, 4 lanes each); only the samples the clamped reads touch are produced
    fine_fft(Sg, 182, z, w400, T, tid, fine_clamp_pos(tb0 - 8 + 32 * 36), fine_clamp_pos(tb0 + 6 + 32 * 42) + 32);
#pragma unroll 1
    for (int r = 0; r < (224 + FINE_NT - 1) / FINE_NT; r++) {
        const int task = tid + FINE_NT * r, qd = task >> 2, n2 = task & 3;
        const bool valid = qd < 56;
        const int ti = valid ? qd / 7 : 0, a = valid ? qd - 7 * ti : 0;
        float mag[8];
        fine_sym_quad<7>(z, tb0 - 8 + 2 * ti + 32 * (36 + a), n2, wq, mag);
        if (valid && n2 == 0) {
            const int c = d_COSTAS[a];
            double off = 0.0, on = 0.0;
#pragma unroll
            for (int b = 0; b < 7; b++) { const double m = (double)mag[b]; on = (b == c) ? m : on; off += (b == c) ? 0.0 : m; }
            dsum[(ti * 7 + a) * 2] = on; dsum[(ti * 7 + a) * 2 + 1] = off;
        }
    }
    __syncthreads();
    int tt = -8; float score_f0 = 0.0f;
    for (int u = 0; u < 8; u++) {                                   // every thread: the same sums in the same order (a ascending, fp64)
        double s1 = 0.0, s2 = 0.0;
        for (int a = 0; a < 7; a++) { s1 += dsum[(u * 7 + a) * 2]; s2 += dsum[(u * 7 + a) * 2 + 1]; }
        const float v = (float)(s1 + W6 * s2);
        if (u == 0 || v > score_f0) { score_f0 = v; tt = -8 + 2 * u; }
    }
    // --- frequency tweaks range(-32, 33, 8): one series each, the block at the chosen time tweak; f = 0 is the time scan's winner
    float best = 0.0f; int ft = 0;
    const int tb = tb0 + tt;
#pragma unroll 1
    for (int i = 0; i < 9; i++) {
        const int f = -32 + 8 * i;
        float v = score_f0;
        if (f != 0) {
            __syncthreads();                                        // everybody is done with the previous series / sums
            fine_fft(Sg, 182 + f, z, w400, T, tid, fine_clamp_pos(tb + 32 * 36), fine_clamp_pos(tb + 32 * 42) + 32);
            {
                const int a = tid >> 2, n2 = tid & 3;
                const bool valid = a < 7;
                float mag[8];
                fine_sym_quad<7>(z, tb + 32 * (36 + (valid ? a : 0)), n2, wq, mag);
                if (valid && n2 == 0) {
                    const int c = d_COSTAS[a];
                    double off = 0.0, on = 0.0;
#pragma unroll
                    for (int b = 0; b < 7; b++) { const double m = (double)mag[b]; on = (b == c) ? m : on; off += (b == c) ? 0.0 : m; }
                    dsum[a * 2] = on; dsum[a * 2 + 1] = off;
                }
            }
            __syncthreads();
            double s1 = 0.0, s2 = 0.0;
            for (int a = 0; a < 7; a++) { s1 += dsum[a * 2]; s2 += dsum[a * 2 + 1]; }
            v = (float)(s1 + W6 * s2);
        }
        if (i == 0 || v > best) { best = v; ft = f; }
    }
    // --- the 79 x 8 grid from the series of the chosen tweaks, symbol by symbol at the clamped positions
    __syncthreads();
    fine_fft(Sg, 182 + ft, z, w400, T, tid, 0, 3200);
#pragma unroll 1
    for (int r = 0; r < (316 + FINE_NT - 1) / FINE_NT; r++) {
        const int task = tid + FINE_NT * r, sy = task >> 2, n2 = task & 3;
        const bool valid = sy < 79;
        float mag[8];
        fine_sym_quad<8>(z, tb + 32 * (valid ? sy : 0), n2, wq, mag);
        if (valid && n2 == 0) {
#pragma unroll
            for (int b = 0; b < 8; b++) mg[sy * 8 + b] = mag[b];
        }
    }
    __syncthreads();
    // --- Costas gate, LLRs, record: as k_fine (receiver.py:164-173)
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
    if (trip && t_sgrid) for (int i = tid; i < 632; i += FINE_NT) t_sgrid[(size_t)bid * 632 + i] = mg[i];
    int ret = 1; float sd = 0.0f; int snr = 0;
    if (nsync <= 6) ret = 0;
    else {
        for (int i = tid; i < 464; i += FINE_NT) p[i] = 20.0f * ft8_log10f(mg[8 * (int)d_PAYSYM[i >> 3] + (i & 7)]);
        __syncthreads();
        llr_from_p(p, llr, sq, tid, tid < 64, &sd, &snr);
        if (tid == 0) { sc[10] = sd; ish[2] = snr; }
        __syncthreads();
        sd = sc[10]; snr = ish[2];
        if (sd <= cfg.llr_sd_min) ret = -1;
        float* out = llr0 + (size_t)bid * 174;
        for (int i = tid; i < 174; i += FINE_NT) out[i] = llr[i];
    }
    if (tid == 0) {
        if (trip) { int32_t* o = t_out + 5 * (size_t)bid; o[0] = ret; o[1] = tt; o[2] = ft; o[3] = nsync; o[4] = snr; t_sd[bid] = sd; }
        else {
            ft8rx_record& r = rec[(size_t)frame * MAXC + ci];
            r.ttweak = (int8_t)tt; r.ftweak = (int8_t)ft; r.nsync = (uint8_t)nsync;
            if (ret == 0) r.status = FT8RX_ST_STOP_COSTAS;
            else { r.fine_sd = sd; r.snr_fine = (int8_t)snr; if (ret < 0) r.status = FT8RX_ST_STOP_FINE_SD; }
        }
    }
}
// test entry (trip != nullptr): one block per triple, triples inside the frequency-domain range are left to k_fine.  Pipeline: blocks
// stride over the fine-sync work list and take the candidates k_fine skips.
__global__ __launch_bounds__(FINE_NT) void k_fine_td(const cpx* __restrict__ spec, ft8rx_record* __restrict__ rec,
                                                     const int32_t* __restrict__ ncand, float* __restrict__ llr0, Tables T, ft8rx_config cfg,
                                                     const int32_t* __restrict__ trip, int32_t* __restrict__ t_out,
                                                     float* __restrict__ t_sd, float* __restrict__ t_sgrid, WorkList work) {
    if (trip) {
        const int h0 = trip[3 * blockIdx.x + 2];
        if (h0 >= FT8RX_MIN_H0_FD && h0 <= FT8RX_MAX_H0_FD) return;
        fine_td_candidate(threadIdx.x, blockIdx.x, spec, rec, ncand, llr0, T, cfg, trip, t_out, t_sd, t_sgrid);
        return;
    }
    const int n = *work.count;
#pragma unroll 1
    for (int item = blockIdx.x; item < n; item += gridDim.x) {
        fine_td_candidate(threadIdx.x, work.items[item], spec, rec, ncand, llr0, T, cfg, nullptr, nullptr, nullptr, nullptr);
        __syncthreads();
    }
}
#endif  // !FT8RX_ILP_UNIT

#endif
