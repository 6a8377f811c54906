// cycle_spectrum.hpp -- 192000-point real FFT of the whole cycle (receiver.py:280-286)
// Part of libft8rx.so; included by ft8rx.hip (single translation unit: the kernels share __constant__/__device__ tables).
#ifndef FT8RX_CYCLE_SPECTRUM_HPP
#define FT8RX_CYCLE_SPECTRUM_HPP

// ------------------------------------------------------------------------------------ cycle spectrum: 192000-pt real FFT
// z[m] = x[2m] + i x[2m+1], 96000 = 300 x 320 four-step, then the real split for bins < 49152.
__global__ __launch_bounds__(256) void k_cyc_a(const int16_t* __restrict__ audio, cpx* __restrict__ A, Tables T) {
    __shared__ cpx bufA[8 * 300];
    __shared__ cpx bufB[8 * 300];
    // XCD-aware tile mapping (workgroup id % 8 = XCD, gridDim.x = 40 = 8 * 5): one XCD takes 5 adjacent column tiles,
    // i.e. 160 contiguous bytes of every audio row, so the 128-B lines are shared inside one L2 instead of four.
    static_assert(320 / 8 == 8 * 5, "the tile map below assumes 40 column tiles = 8 XCDs x 5");
    const int tile = (blockIdx.x & 7) * 5 + (blockIdx.x >> 3);
    const int f = blockIdx.y, tid = threadIdx.x, n2b = 8 * tile;
    const int16_t* a = audio + (size_t)f * FT8RX_NSAMP;
    for (int i = tid; i < 2400; i += 256) {
        int c = i & 7, n1 = i >> 3;
        int m = 320 * n1 + n2b + c;
        float re = 0.0f, im = 0.0f;
        if (2 * m < FT8RX_NSAMP) { short2 v = *reinterpret_cast<const short2*>(a + 2 * m); re = (float)v.x; im = (float)v.y; }
        bufA[c * 300 + n1] = make_float2(re, im);
    }
    __syncthreads();
    cpx* r = lds_fft<300, 5, 5, 4, 3>(bufA, bufB, T.W300, 8, tid, 256);
    cpx* out = A + (size_t)f * 96000;
    for (int i = tid; i < 2400; i += 256) {
        int c = i & 7, k1 = i >> 3, n2 = n2b + c;
        cpx v = r[c * 300 + k1];
        if (n2 * k1 != 0) v = cmul(v, T.W96000[n2 * k1]);
        out[k1 * 320 + n2] = v;
    }
}

__global__ __launch_bounds__(256) void k_cyc_b(const cpx* __restrict__ A, cpx* __restrict__ Z, Tables T) {
    __shared__ cpx bufA[4 * 320];
    __shared__ cpx bufB[4 * 320];
    const int f = blockIdx.y, tid = threadIdx.x, k1b = 4 * blockIdx.x;
    const cpx* in = A + (size_t)f * 96000 + (size_t)k1b * 320;
    for (int i = tid; i < 1280; i += 256) bufA[i] = in[i];
    __syncthreads();
    cpx* r = lds_fft<320, 8, 8, 5>(bufA, bufB, T.W320, 4, tid, 256);
    cpx* out = Z + (size_t)f * 96000;
    for (int i = tid; i < 1280; i += 256) {
        int rr = i & 3, k2 = i >> 2;
        out[(k1b + rr) + 300 * k2] = r[rr * 320 + k2];
    }
}

__global__ __launch_bounds__(256) void k_cyc_c(const cpx* __restrict__ Z, cpx* __restrict__ spec, Tables T) {
    const int f = blockIdx.y, k = blockIdx.x * 256 + threadIdx.x;
    const cpx* z = Z + (size_t)f * 96000;
    cpx p = z[k], q = z[(96000 - k) % 96000];
    float er = 0.5f * (p.x + q.x), ei = 0.5f * (p.y - q.y);
    float orr = 0.5f * (p.y + q.y), oi = 0.5f * (q.x - p.x);
    cpx w = T.WR192k[k];
    spec[(size_t)f * FT8RX_SPEC_BINS + k] = make_float2(er + (w.x * orr - w.y * oi), ei + (w.x * oi + w.y * orr));
}

#endif
