// bp.hpp -- LDPC(174,91) belief propagation and the ladder select kernels (decoders.py:140-171, receiver.py:68-107)
// Part of libft8rx.so; included by ft8rx.hip (single translation unit: the kernels share __constant__/__device__ tables).
#ifndef FT8RX_BP_HPP
#define FT8RX_BP_HPP

// ------------------------------------------------------------------------------------ LDPC belief propagation
// One wavefront per (candidate, AP) -- or per test vector.  Edge-parallel tanh / message update
// (lane l owns edges l, l+64, ...), check-parallel products, variable-parallel accumulation in the
// reference's np.add.at order; everything exchanged through LDS; parity via __ballot.
// mode 0: pipeline ipass 0, BP(nc0_a, iters_a) of the attempts that bp0_precheck (llr.hpp: GOOD91 + initial check count, done
// where the grid LLRs are produced) left pending; mode 1: pipeline fine stage, BP(nc0_b, iters_b) of one AP variant with its
// output llr saved -- the ap 0 attempt first evaluates GOOD91 for ap 0 and ap 1 (ipass 2) and skips its BP if either succeeds;
// mode 2: raw vectors (tests).
// The fine-stage attempts run in ladder order in three launches (receiver.py:84-98: GOOD91 ap 0,1; BP_A ap 0,1; BP_B ap 0..4, first
// success wins): {GOOD91 x 2, BP ap 0} -> k_select1(0) -> {BP ap 1} -> k_select1(1) -> {BP ap 2,3,4} -> k_select1(2).  A candidate
// that is decided leaves the lists; with all five variants in one launch 42 % of the candidates (the ones that decode here) ran
// four BPs nobody reads, most of them 20 iterations on wrongly forced bits (profiles/archive/r02_notes.md).
#ifndef BP_WV
#define BP_WV 7          /* <= 72 VGPRs: 7 waves per SIMD; measured 1.76 -> 1.68 ms for both BP launches (profiles/archive/r02_notes.md) */
#endif
// Timing-only builds (-DBP_TIMING, tools/bp_timing.py): lane 0 of every attempt adds the shader cycles between consecutive marks to a
// per-block slot of g_bp_t[] (spread over 65536 slots: the atomics do not meet).  Never defined in the product.
#ifdef BP_TIMING
__device__ unsigned long long g_bp_t[65536][8];
#define BT_DECL unsigned long long bt_prev = __builtin_readcyclecounter();
#define BT(i) do { const unsigned long long bt_now = __builtin_readcyclecounter(); if (lane == 0) atomicAdd(&g_bp_t[blockIdx.x & 65535][i], bt_now - bt_prev); bt_prev = __builtin_readcyclecounter(); } while (0)
#else
#define BT_DECL
#define BT(i) do { } while (0)
#endif
FT8_DEV void bp_attempt(int lane, int mode, int bid, const float* __restrict__ llr_in, ft8rx_record* __restrict__ rec,
                        const int32_t* __restrict__ ncand, Att* __restrict__ attG, Att* __restrict__ attB,
                        float* __restrict__ saved, ft8rx_event* ev, int32_t* evcount, const ft8rx_config& cfg,
                        int max_nc0, int max_iters) {
    __shared__ float llr[176];
    __shared__ float tl[576];        // 9 x 64 edge slots: slots >= 522 are dummy edges (variable 174, check 83) so that the
    // per-edge code below is straight-line for all nine slots of a lane.  The message deltas overwrite the tanh values in place: a
    // lane reads tl[e] and writes dl[e] of its OWN slots only, after the barrier behind the check products (the only readers of other
    // lanes' tanh values) -- 2.3 KB of LDS less per wave: bp_fine 0.719 -> 0.694 ms, bp_grid 0.171 -> 0.167 (profiles/archive/r03_notes.md)
    float* dl = tl;
    __shared__ float P[84];
    int frame = 0, ci = 0, ap = 0; size_t vec;
    BT_DECL
    if (mode == 2) vec = bid;
    else {
        ap = bid % 5; int c = bid / 5; frame = c / MAXC; ci = c % MAXC;
        if (ci >= ncand[frame]) return;
        if (rec[(size_t)frame * MAXC + ci].status != FT8RX_ST_ACTIVE) return;
        vec = (size_t)c;
    }
    {   // three loads in flight, then the AP override (ap_value around the load would branch over it: one round trip per basic block)
        const float* src = llr_in + vec * 174;
        const float v0 = src[lane], v1 = src[64 + lane], v2 = src[128 + (lane < 46 ? lane : 0)];
        llr[lane] = ap_value(ap, lane, v0); llr[64 + lane] = ap_value(ap, 64 + lane, v1);
        if (lane < 46) llr[128 + lane] = ap_value(ap, 128 + lane, v2);
    }
    if (lane < 2) llr[174 + lane] = 0.0f;
    if (lane == 0) P[83] = 1.0f;
    __syncthreads();
    Att res; memset(&res, 0, sizeof(res)); res.n_its = -1;
    // ---- GOOD91 of the fine stage (ipass 2, ap 0 then ap 1): CRC on the hard decisions of llr[:91] (receiver.py:119-122)
    // (ipass 0's GOOD91 is done by bp0_precheck in k_grid_llr)
    if (mode == 1 && ap == 0) {
        bool any = false;
#pragma unroll
        for (int g = 0; g < 2; g++) {
            Att resG; memset(&resG, 0, sizeof(resG)); resG.n_its = -1;
            const uint64_t b0 = __ballot(ap_value(g, lane, llr[lane]) > 0.0f);
            const uint64_t b1 = __ballot(lane < 27 && ap_value(g, 64 + lane, llr[64 + (lane < 27 ? lane : 0)]) > 0.0f);
            uint64_t lo, hi;
            const int r = ft8_crc_check_wave(b0, b1, lane, &lo, &hi);
            if (r) { if (lane == 0) log_event(ev, evcount, frame, ci, 2, g, 0, lo, hi, r == 2); }
            if (r == 2) { resG.ok = 1; resG.lo = lo; resG.hi = hi; resG.n_its = 0; resG.method = FT8RX_M_GOOD91; any = true; }
            if (lane == 0) attG[vec * 2 + g] = resG;
        }
        if (any) return;                       // decided at ipass 2: k_select1(0) never looks at this candidate's BP records
    }
    // membership masks of this lane's two checks (c0 = lane, c1 = 64 + lane) over the 174 variables
    const int c0 = lane, c1 = lane + 64;
    const uint64_t cm00 = d_CHK_MASK[c0][0], cm01 = d_CHK_MASK[c0][1], cm02 = d_CHK_MASK[c0][2];
    const uint64_t cm10 = d_CHK_MASK[c1][0], cm11 = d_CHK_MASK[c1][1], cm12 = d_CHK_MASK[c1][2];
    // the edge tables are only needed once BP really iterates: most ipass-0 attempts stop at the initial
    // unsatisfied-check test (decoders.py:159), so they are loaded lazily below
    int ev_[9], ec_[9];
    uint32_t ve_[3] = {0u, 0u, 0u};          // the three edges of this lane's variables (lane, 64 + lane, 128 + lane), 10 bits each
    int n0 = 0, n1 = 0;
    bool tables = false;
    float mc[9];
#pragma unroll
    for (int i = 0; i < 9; i++) mc[i] = 0.0f;
    res.has_out = 1;
    BT(0);
    for (int it = 0; it < max_iters; it++) {
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
            int r = ft8_crc_check_wave(b0, b1, lane, &lo, &hi);
            if (r) {
                int ipass = (mode == 0) ? 0 : ((ap < 2 && res.nc0 <= cfg.bp_nc0_a && it < cfg.bp_iters_a) ? 3 : 4);
                if (lane == 0) log_event(ev, evcount, frame, ci, ipass, ap, it + 1, lo, hi, r == 2);
            }
            if (r == 2) { res.ok = 1; res.lo = lo; res.hi = hi; res.n_its = (int16_t)it; res.has_out = 0; }
            break;      // success, or frozen state: the reference changes nothing from here on (decoders.py:161-164)
        }
        BT(1);
        if (!tables) {                 // wave-uniform: first real iteration
            tables = true;
#pragma unroll
            for (int i = 0; i < 9; i++) { int e = lane + 64 * i; ev_[i] = (e < 522) ? d_EDGE_V[e] : 174; ec_[i] = (e < 522) ? d_EDGE_C[e] : 83; }
            n0 = d_CHK_N[c0];
            n1 = (c1 < 83) ? d_CHK_N[c1] : 0;
            // (these used to be read from the global table in EVERY iteration: three dependent global loads in front of the LDS reads)
#pragma unroll
            for (int q = 0; q < 3; q++) {
                const int v = lane + 64 * q, vc = v < 174 ? v : 0;
                ve_[q] = (uint32_t)d_VAR_E[vc][0] | ((uint32_t)d_VAR_E[vc][1] << 10) | ((uint32_t)d_VAR_E[vc][2] << 20);
            }
        }
#pragma unroll
        for (int i = 0; i < 9; i++) {
            const int e = lane + 64 * i;
            const float v2c = llr[ev_[i]] - mc[i];
            tl[e] = ft8_tanhf(-v2c);
        }
        BT(2);
        __syncthreads();
        {
            // edge j of check c sits in slot 83 j + c, the seventh edge of a degree-7 check (c >= 59) in slot 498 + c - 59
            // (ft8rx_create): consecutive lanes read consecutive words
            float Pp = tl[c0];
#pragma unroll
            for (int j = 1; j < 6; j++) Pp = Pp * tl[c0 + 83 * j];
            if (n0 == 7) Pp = Pp * tl[439 + c0];
            P[c0] = Pp;
            if (c1 < 83) {
                float Q = tl[c1];
#pragma unroll
                for (int j = 1; j < 6; j++) Q = Q * tl[c1 + 83 * j];
                if (n1 == 7) Q = Q * tl[439 + c1];
                P[c1] = Q;
            }
        }
        BT(3);
        __syncthreads();
#pragma unroll
        for (int i = 0; i < 9; i++) {
            const int e = lane + 64 * i;
            const float tti = tl[e];          // own slot, re-read: keeping tt[] in registers across the barriers spilled 24 B per thread at 72 VGPRs
            const float Pc = P[ec_[i]];
            const float nm = (Pc * tti) / (__builtin_fmaf(-1.18f, tti, Pc) * __builtin_fmaf(1.18f, tti, Pc));     // = e/((e-1.18)(1.18+e)), e = P/t, in one division
            dl[e] = nm - mc[i];
            mc[i] = nm;
        }
        BT(4);
        __syncthreads();
#pragma unroll
        for (int q = 0; q < 3; q++) {
            const int v = lane + 64 * q;
            if (v < 174) {
                float col = 0.0f;
                col += dl[ve_[q] & 1023u]; col += dl[(ve_[q] >> 10) & 1023u]; col += dl[ve_[q] >> 20];
                llr[v] += col;
            }
        }
        __syncthreads();
        BT(5);
    }
    BT(6);
    if (res.ok) res.method = (mode == 0) ? FT8RX_M_LDPC_A : FT8RX_M_LDPC_B;
    if (mode == 2) {
        if (lane == 0) attB[vec] = res;
        if (res.has_out) for (int i = lane; i < 174; i += 64) saved[vec * 174 + i] = llr[i];
        return;
    }
    if (mode == 0) { if (lane == 0) attB[vec * 5 + ap] = res; return; }
    if (lane == 0) attB[vec * 5 + ap] = res;
#ifndef BP_TIMING_NO_SAVED
    if (res.has_out) for (int i = lane; i < 174; i += 64) saved[(vec * 5 + ap) * 174 + i] = llr[i];
#endif
}

// mode 2 (test entry): one block per vector.  Pipeline modes: one attempt per block from the work list (BP attempts are short and
// very uneven: the hardware's block dispatcher balances them better than a strided loop, and the straight-line kernel allocates
// registers better); blocks beyond the list exit after one load.
// mode 0: the list holds attempts (candidate * 5 + ap) that survived bp0_precheck; mode 1: candidates, AP variants ap_lo .. ap_lo + ap_n - 1
__global__ __launch_bounds__(64, BP_WV) void k_bp(int mode, const float* __restrict__ llr_in, ft8rx_record* __restrict__ rec,
                                           const int32_t* __restrict__ ncand, Att* __restrict__ attG, Att* __restrict__ attB,
                                           float* __restrict__ saved, ft8rx_event* ev, int32_t* evcount, ft8rx_config cfg,
                                           int max_nc0, int max_iters, WorkList work, int ap_lo, int ap_n) {
    if (mode == 2) { bp_attempt(threadIdx.x, 2, blockIdx.x, llr_in, rec, ncand, attG, attB, saved, ev, evcount, cfg, max_nc0, max_iters); return; }
    // ipass 0: bounded grid (ladder_grid).  The host does not know the list's length, and the grid sized for the worst case (every
    // candidate x 5 variants: 327 k blocks per 256 frames for ~25 k pending attempts) spent a quarter of the launch dispatching
    // blocks that load the count and exit: 0.158 -> 0.12 ms.  With the cap nearly every block still runs at most one attempt.
    if (mode == 0) {
        const int n = *work.count;
#pragma unroll 1
        for (int item = blockIdx.x; item < n; item += gridDim.x) {
            int tid = threadIdx.x;
            asm volatile("" : "+v"(tid));                           // opaque per attempt: nothing thread-specific is carried across attempts (else 100 B of scratch)
            bp_attempt(tid, 0, work.items[item], llr_in, rec, ncand, attG, attB, saved, ev, evcount, cfg, max_nc0, max_iters);
            __syncthreads();                                        // the LDS arrays are reused by the next attempt
        }
        return;
    }
    // fine stage: one attempt per block, blocks beyond the list exit after one load (the looped form measured 4 % slower here: the
    // lists are a sixth to a half of the worst case, and the loop costs the attempt's code more than the empty blocks cost the launch)
    const int item = blockIdx.x;
    if (item >= *work.count * ap_n) return;
    bp_attempt(threadIdx.x, mode, work.items[item / ap_n] * 5 + ap_lo + item % ap_n, llr_in, rec, ncand, attG, attB, saved, ev, evcount, cfg, max_nc0, max_iters);
}

// first success in ladder order after ipass 0 (receiver.py:72-78)
__global__ void k_select0(ft8rx_record* rec, const int32_t* ncand, const Att* att0, int B, WorkList next) {
    int c = blockIdx.x * blockDim.x + threadIdx.x;
    bool go_on = false;
    if (c < B * MAXC && (c % MAXC) < ncand[c / MAXC]) {
        ft8rx_record& r = rec[c];
        if (r.status == FT8RX_ST_ACTIVE) {
            go_on = true;
            for (int ap = 0; ap < 5; ap++) {
                const Att& a = att0[(size_t)c * 5 + ap];
                if (a.ok) { r.status = FT8RX_ST_DECODED; r.ipass = 0; r.ap = (uint8_t)ap; r.method = a.method; r.n_its = a.n_its; r.msg_lo = a.lo; r.msg_hi = a.hi; go_on = false; break; }
            }
        }
    }
    work_push_block(next, go_on, c);                                // still undecoded: goes on to the fine sync
}

// work list of a ladder step = every candidate that is ACTIVE now (thread per candidate, one atomic per block)
__global__ void k_worklist(const ft8rx_record* rec, const int32_t* ncand, int B, WorkList next) {
    int c = blockIdx.x * blockDim.x + threadIdx.x;
    const bool on = c < B * MAXC && (c % MAXC) < ncand[c / MAXC] && rec[c].status == FT8RX_ST_ACTIVE;
    work_push_block(next, on, c);
}

// attempt list of the first BP: thread per (candidate, ap); pending = left open by bp0_precheck
__global__ void k_worklist_att(const ft8rx_record* rec, const int32_t* ncand, const Att* att0, int B, WorkList next) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x, c = i / 5;
    const bool on = c < B * MAXC && (c % MAXC) < ncand[c / MAXC] && rec[c].status == FT8RX_ST_ACTIVE && att0[i].pad[1] != 0;
    work_push_block(next, on, i);
}

// Fine-stage ladder (receiver.py:84-98), order: GOOD91 ap 0, ap 1 (ipass 2); BP_A ap 0, ap 1 (ipass 3: the BP_B run of that variant
// if it would also have succeeded under BP_A's limits); BP_B ap 0..4 (ipass 4).  step 0 runs after {GOOD91 x 2, BP ap 0}, step 1
// after {BP ap 1}, step 2 after {BP ap 2,3,4}; a step decides what the attempts so far can decide and passes the rest on.
// step 3 = the whole ladder at once, after a single launch of all five variants (small batches: three dependent launches of up
// to 20 iterations each triple the latency of this stage when there are too few attempts to fill the GPU anyway).
FT8_DEV bool sel_take(ft8rx_record& r, const Att& a, int ipass, int ap, int method) {
    r.status = FT8RX_ST_DECODED; r.ipass = (uint8_t)ipass; r.ap = (uint8_t)ap; r.method = (uint8_t)method; r.n_its = a.n_its; r.msg_lo = a.lo; r.msg_hi = a.hi;
    return false;
}
__global__ void k_select1(int step, ft8rx_record* rec, const int32_t* ncand, const Att* attG, const Att* attB, int B, ft8rx_config cfg, WorkList next) {
    int c = blockIdx.x * blockDim.x + threadIdx.x;
    bool go_on = false;
    if (c < B * MAXC && (c % MAXC) < ncand[c / MAXC] && rec[c].status == FT8RX_ST_ACTIVE) {
        ft8rx_record& r = rec[c];
        go_on = true;
        const Att* b = attB + (size_t)c * 5;
#define IS_A(a) ((a).ok && (a).nc0 <= cfg.bp_nc0_a && (a).n_its < cfg.bp_iters_a)
        if (step == 0) {
            const Att& g0 = attG[(size_t)c * 2], & g1 = attG[(size_t)c * 2 + 1];
            if (g0.ok) go_on = sel_take(r, g0, 2, 0, FT8RX_M_GOOD91);
            else if (g1.ok) go_on = sel_take(r, g1, 2, 1, FT8RX_M_GOOD91);
            else if (IS_A(b[0])) go_on = sel_take(r, b[0], 3, 0, FT8RX_M_LDPC_A);
        } else if (step == 1) {
            if (IS_A(b[1])) go_on = sel_take(r, b[1], 3, 1, FT8RX_M_LDPC_A);
            else if (b[0].ok) go_on = sel_take(r, b[0], 4, 0, FT8RX_M_LDPC_B);
            else if (b[1].ok) go_on = sel_take(r, b[1], 4, 1, FT8RX_M_LDPC_B);
        } else if (step == 2) {
            for (int ap = 2; ap < 5 && go_on; ap++) if (b[ap].ok) go_on = sel_take(r, b[ap], 4, ap, FT8RX_M_LDPC_B);
        } else {                                                    // step 3: all five variants ran in one launch (small batches)
            const Att& g0 = attG[(size_t)c * 2], & g1 = attG[(size_t)c * 2 + 1];
            if (g0.ok) go_on = sel_take(r, g0, 2, 0, FT8RX_M_GOOD91);
            else if (g1.ok) go_on = sel_take(r, g1, 2, 1, FT8RX_M_GOOD91);
            else if (IS_A(b[0])) go_on = sel_take(r, b[0], 3, 0, FT8RX_M_LDPC_A);
            else if (IS_A(b[1])) go_on = sel_take(r, b[1], 3, 1, FT8RX_M_LDPC_A);
            else for (int ap = 0; ap < 5 && go_on; ap++) if (b[ap].ok) go_on = sel_take(r, b[ap], 4, ap, FT8RX_M_LDPC_B);
        }
#undef IS_A
    }
    work_push_block(next, go_on, c);                                // still undecoded: next BP step, or OSD after step 2
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
            r.method = (s < 5) ? FT8RX_M_OSD : FT8RX_M_LDPC_B_OSD; r.n_its = a.n_its; r.msg_lo = a.lo; r.msg_hi = a.hi; r.osd_hd = a.pad[0]; return; }
    }
    r.status = FT8RX_ST_EXHAUSTED;
}

#endif
