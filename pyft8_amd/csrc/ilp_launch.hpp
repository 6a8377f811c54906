// ilp_launch.hpp -- launchers of the kernels that live in the second translation unit (ft8rx_ilp.hip); included by both units
#ifndef FT8RX_ILP_LAUNCH_HPP
#define FT8RX_ILP_LAUNCH_HPP
void ft8rx_ilp_spectrogram(int n_frames, hipStream_t s, const int16_t* audio, float* grid, const Tables& T);
void ft8rx_ilp_hop_spectrum(hipStream_t s, const int16_t* win3840, float* row, const Tables& T);
void ft8rx_ilp_fine(int n_blocks, hipStream_t s, const cpx* spec, ft8rx_record* rec, const int32_t* ncand, float* llr0, const Tables& T,
                    const ft8rx_config& cfg, const int32_t* trip, int32_t* t_out, float* t_sd, float* t_sgrid, WorkList work);
#ifdef FINE_TIMING
int ft8rx_ilp_fine_times(unsigned long long* out32, int reset);
#endif
#endif
