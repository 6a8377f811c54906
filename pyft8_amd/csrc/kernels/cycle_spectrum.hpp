// cycle_spectrum.hpp -- 192000-point real FFT of the whole cycle (receiver.py:280-286)
// Part of libft8rx.so; included by ft8rx.hip (single translation unit: the kernels share __constant__/__device__ tables).
#ifndef FT8RX_CYCLE_SPECTRUM_HPP
#define FT8RX_CYCLE_SPECTRUM_HPP

// ------------------------------------------------------------------------------------ cycle spectrum: 192000-pt real FFT
// z[m] = x[2m] + i x[2m+1], 96000 = 300 x 320 four-step, then the real split for bins < 49152.
// The 10 raw samples a thread contributes are requested up front (straight-line, from clamped addresses), then converted into the
// LDS tile: the former load->store loop waited for every load separately (191 -> 146 us; two to four frames per block with the
// later frames' samples prefetched behind the first transform gain nothing on top: 142 us).
FT8_DEV void cyc_a_load(const int16_t* __restrict__ a, int n2b, int tid, uint32_t (&raw)[10]) {
#pragma unroll
    for (int q = 0; q < 10; q++) {
        const int i = tid + 256 * q, ic = i < 2400 ? i : 0;
        const int c = ic & 7, n1 = ic >> 3;
        const int m = 320 * n1 + n2b + c;
        // zero padding beyond the 180000 samples (receiver.py:283: a 192000-sample buffer) by integer masking of a clamped load -- a
        // select would be turned back into a branch around the load (one memory round trip per basic block)
        const int inb = (2 * m < FT8RX_NSAMP) ? -1 : 0;
        raw[q] = *reinterpret_cast<const uint32_t*>(a + ((2 * m) & inb)) & (uint32_t)inb;
    }
}
FT8_DEV void cyc_a_frame(const uint32_t (&raw)[10], cpx* bufA, cpx* bufB, cpx* __restrict__ out, const Tables& T, int n2b, int tid) {
#pragma unroll
    for (int q = 0; q < 10; q++) {
        const int i = tid + 256 * q;
        if (i < 2400) bufA[(i & 7) * 300 + (i >> 3)] = make_float2((float)(int16_t)(raw[q] & 0xFFFFu), (float)(int16_t)(raw[q] >> 16));
    }
    // the four-step twiddles of this thread's ten outputs: requested before the transform, used after it
    cpx w[10];
#pragma unroll
    for (int q = 0; q < 10; q++) { const int i = tid + 256 * q, ic = i < 2400 ? i : 0; w[q] = T.W96000[(n2b + (ic & 7)) * (ic >> 3)]; }
    __syncthreads();
    cpx* r = lds_fft<300, 5, 5, 4, 3>(bufA, bufB, T.W300, 8, tid, 256);
#pragma unroll
    for (int q = 0; q < 10; q++) {
        const int i = tid + 256 * q;
        if (i < 2400) { const int c = i & 7, k1 = i >> 3; out[k1 * 320 + n2b + c] = cmul(r[c * 300 + k1], w[q]); }   // unconditional: W^0 = (1, -0) is an exact identity
    }
}
__global__ __launch_bounds__(256) void k_cyc_a(const int16_t* __restrict__ audio, cpx* __restrict__ A, Tables T) {
    __shared__ cpx bufA[8 * 300];
    __shared__ cpx bufB[8 * 300];
    // XCD-aware tile mapping (workgroup id % 8 = XCD, gridDim.x = 40 = 8 * 5): one XCD takes 5 adjacent column tiles,
    // i.e. 160 contiguous bytes of every audio row, so the 128-B lines are shared inside one L2 instead of four.
    static_assert(320 / 8 == 8 * 5, "the tile map below assumes 40 column tiles = 8 XCDs x 5");
    const int tile = (blockIdx.x & 7) * 5 + (blockIdx.x >> 3);
    const int f = blockIdx.y, tid = threadIdx.x, n2b = 8 * tile;
    uint32_t raw[10];
    cyc_a_load(audio + (size_t)f * FT8RX_NSAMP, n2b, tid, raw);
    cyc_a_frame(raw, bufA, bufB, A + (size_t)f * 96000, T, n2b, tid);
}

// Row FFTs (320-point, over n2) fused with the real-FFT split.  Z[k1 + 300 k2] = FFT320 of row k1 of A; the split of bin k needs
// Z[k] and Z[96000 - k], and 96000 - (k1 + 300 k2) = (300 - k1) + 300 (319 - k2): the partner of row k1 is row 300 - k1.  A block
// therefore transforms 4 adjacent rows k1 .. k1+3 TOGETHER WITH their partner rows 297-k1 .. 300-k1 and forms the spectrum bins of
// all eight rows from LDS -- the 96000-point intermediate Z never goes to HBM (r01: 197 MB written + 201 MB read per 256 frames).
// Blocks 0..37: k1 = 1 + 4 b (rows 1..152 and 148..299; the rows 148..152 are produced twice, identically); block 38: the
// self-paired rows 0 (partner bin 300 (320 - k2)) and 150.  Only bins < FT8RX_SPEC_BINS are kept (k2 < 164; all 320 in the wide build).
// XCD-aware tile map as in k_cyc_a (gridDim.x = 40 = 8 XCDs x 5, workgroup id % 8 = XCD): one XCD takes five adjacent row
// blocks = 20 adjacent rows, so the 32-byte output runs of neighbouring blocks (bin k = row + 300 k2) meet in ONE L2 and leave it
// as whole lines; with the plain map the partial lines went out from four L2s (WRITE_SIZE 156 MB for 101 MB of spectrum).
#define CYC_BC_BLOCKS 39
#define CYC_BC_GRID 40
FT8_DEV int cyc_bc_row(int b, int r) {             // row index of slot r (0..7): slots 0..3 = the low rows, 4..7 = their partners
    if (b == CYC_BC_BLOCKS - 1) return (r == 0) ? 0 : 150;      // special block: slot 0 = row 0, every other slot = row 150 (only slot 1 is used)
    const int k1a = 1 + 4 * b;
    return (r < 4) ? k1a + r : 300 - (k1a + (r - 4));
}
FT8_DEV void cyc_bc_load(const cpx* __restrict__ in, int b, int tid, cpx (&v)[10]) {
#pragma unroll
    for (int q = 0; q < 10; q++) {
        const int i = tid + 256 * q, r = i / 320, c = i - 320 * r;
        v[q] = in[(size_t)cyc_bc_row(b, r) * 320 + c];
    }
}
FT8_DEV void cyc_bc_frame(const cpx (&v)[10], cpx* bufA, cpx* bufB, cpx* __restrict__ out, const Tables& T, int b, int tid) {
#pragma unroll
    for (int q = 0; q < 10; q++) bufA[tid + 256 * q] = v[q];
    __syncthreads();
    cpx* z = lds_fft<320, 8, 8, 5>(bufA, bufB, T.W320, 8, tid, 256);
    const bool special = (b == CYC_BC_BLOCKS - 1);
    // outputs: slot r, k2 < NK2 -> bin k = row + 300 k2; adjacent threads take adjacent rows (32-byte runs in memory).
    // (requesting the split twiddles before the transform, as k_cyc_a does with its four-step twiddles, is slower here: 101 -> 111 us)
    constexpr int NK2 = (FT8RX_SPEC_BINS + 299) / 300;                          // 164 (bins < 49152), 320 in the wide build
    static_assert(NK2 <= 320, "k2 range of the 320-point row transforms");
    for (int i = tid; i < 8 * NK2; i += 256) {
        const int half = i / (4 * NK2), j = i - half * (4 * NK2), rr = j & 3, k2 = j >> 2;
        const int r = 4 * half + rr;
        int pr, pk2;                                                            // partner slot, partner k2
        if (special) {
            if (r >= 2) continue;
            pr = r;                                                             // self-paired
            pk2 = (r == 0) ? ((320 - k2) % 320) : 319 - k2;
        } else {
            pr = (r < 4) ? r + 4 : r - 4;
            pk2 = 319 - k2;
        }
        const int k = cyc_bc_row(b, r) + 300 * k2;
        if (k >= FT8RX_SPEC_BINS) continue;
        const cpx p = z[r * 320 + k2], q = z[pr * 320 + pk2];
        const float er = 0.5f * (p.x + q.x), ei = 0.5f * (p.y - q.y);
        const float orr = 0.5f * (p.y + q.y), oi = 0.5f * (q.x - p.x);
        const cpx w = T.WR192k[k];
        out[k] = make_float2(er + (w.x * orr - w.y * oi), ei + (w.x * oi + w.y * orr));
    }
}
__global__ __launch_bounds__(256) void k_cyc_bc(const cpx* __restrict__ A, cpx* __restrict__ spec, Tables T) {
    __shared__ cpx bufA[8 * 320];
    __shared__ cpx bufB[8 * 320];
    static_assert(CYC_BC_GRID == 8 * 5 && CYC_BC_BLOCKS <= CYC_BC_GRID, "tile map: 40 slots = 8 XCDs x 5");
    const int f = blockIdx.y, tid = threadIdx.x, b = (blockIdx.x & 7) * 5 + (blockIdx.x >> 3);
    if (b >= CYC_BC_BLOCKS) return;
    cpx v[10];                                    // the thread's ten input points, requested together (151 -> 99 us against the load->store loop)
    cyc_bc_load(A + (size_t)f * 96000, b, tid, v);
    cyc_bc_frame(v, bufA, bufB, spec + (size_t)f * FT8RX_SPEC_BINS, T, b, tid);
}

#endif
