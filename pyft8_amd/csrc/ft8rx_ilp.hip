// ft8rx_ilp.hip -- second translation unit of libft8rx.so: the two FFT kernels (k_fine, k_spectrogram / k_hop_spectrum), compiled
// with a different instruction-scheduling strategy than the rest of the library.
//
// hipcc's default scheduler and `-mllvm -amdgpu-sched-strategy=iterative-ilp` produce the same arithmetic in a different order; on
// gfx950 the ILP strategy is 3.9 % faster for k_fine (2.675 -> 2.570 ms per 256 frames) and 3 % for k_spectrogram, but 56 % SLOWER
// for k_bp (profiles/archive/r03_notes.md) -- and the strategy can only be chosen per translation unit.  So these kernels are built here,
// from the same headers as ft8rx.hip, and ft8rx.hip launches them through the three functions at the bottom.  FT8RX_ILP_UNIT selects
// this unit's share of the headers: the FFT device code, the three kernels and the plain structs / constant tables they need -- no
// other kernel and none of the run-time initialised device tables (those live in the main unit only), so every kernel exists exactly
// once in libft8rx.so.  Results are bit-identical with either scheduler: the GPU parity suite runs against this build.
#define FT8RX_ILP_UNIT 1
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <string.h>
#include "../../include/ft8rx.h"
#include "ft8_dev.h"

#define MAXC FT8RX_MAX_CANDS
#define NF0MAX (FT8RX_MAX_F0 > 1024 ? 2048 : 1024)

#include "kernels/common.hpp"
#include "kernels/spectrogram.hpp"
#include "kernels/llr.hpp"
#include "kernels/fine_sync.hpp"
#include "ilp_launch.hpp"

void ft8rx_ilp_spectrogram(int n_frames, hipStream_t s, const int16_t* audio, float* grid, const Tables& T) {
    k_spectrogram<<<dim3(376, n_frames), SPEC_NT, 0, s>>>(audio, grid, T);
}
void ft8rx_ilp_hop_spectrum(hipStream_t s, const int16_t* win3840, float* row, const Tables& T) {
    k_hop_spectrum<<<1, SPEC_NT, 0, s>>>(win3840, row, T);
}
void ft8rx_ilp_fine(int n_blocks, hipStream_t s, const cpx* spec, ft8rx_record* rec, const int32_t* ncand, float* llr0, const Tables& T,
                    const ft8rx_config& cfg, const int32_t* trip, int32_t* t_out, float* t_sd, float* t_sgrid, WorkList work) {
    k_fine<<<n_blocks, FINE_NT, 0, s>>>(spec, rec, ncand, llr0, T, cfg, trip, t_out, t_sd, t_sgrid, work);
}
#ifdef FINE_TIMING
int ft8rx_ilp_fine_times(unsigned long long* out32, int reset) {
    if (hipDeviceSynchronize() != hipSuccess) return -2;
    if (out32 && hipMemcpyFromSymbol(out32, HIP_SYMBOL(g_fine_t), sizeof(unsigned long long) * 32) != hipSuccess) return -2;
    if (reset) { unsigned long long z[32] = {0}; if (hipMemcpyToSymbol(HIP_SYMBOL(g_fine_t), z, sizeof(z)) != hipSuccess) return -2; }
    return 0;
}
#endif
