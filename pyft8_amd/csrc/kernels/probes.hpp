// probes.hpp -- small probe kernels used by the parity tests
// Part of libft8rx.so; included by ft8rx.hip (single translation unit: the kernels share __constant__/__device__ tables).
#ifndef FT8RX_PROBES_HPP
#define FT8RX_PROBES_HPP

// ------------------------------------------------------------------------------------ small probes (tests)
__global__ void k_math_probe(int which, const float* x, float* y, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) y[i] = (which == 0) ? ft8_log10f(x[i]) : ft8_tanhf(x[i]);
}
template <int N, int... Rs>
__global__ void k_fft_probe(const cpx* x, cpx* y, const cpx* W) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    cpx* a = reinterpret_cast<cpx*>(smem); cpx* b = a + N;
    for (int i = threadIdx.x; i < N; i += blockDim.x) a[i] = x[i];
    __syncthreads();
    cpx* r = lds_fft<N, Rs...>(a, b, W, 1, threadIdx.x, blockDim.x);
    for (int i = threadIdx.x; i < N; i += blockDim.x) y[i] = r[i];
}
__global__ void k_crc_probe(const float* cw91, int n, int32_t* res, uint64_t* lo, uint64_t* hi) {
    int v = blockIdx.x; int lane = threadIdx.x;
    const float* c = cw91 + (size_t)v * 91;
    uint64_t b0 = __ballot(c[lane] > 0.0f);
    uint64_t b1 = __ballot(lane < 27 && c[64 + (lane < 27 ? lane : 0)] > 0.0f);
    uint64_t l, h; int r = ft8_crc_check(b0, b1, &l, &h);
    if (lane == 0) { res[v] = r; lo[v] = l; hi[v] = h; }
}
__global__ void k_valid_probe(const uint64_t* lo, const uint64_t* hi, int n, int32_t* out) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = ft8_valid77(lo[i], hi[i]) ? 1 : 0;
}

#endif
