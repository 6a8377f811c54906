/* ft8_oracle.h -- public types of the CPU oracle.  TEST INFRASTRUCTURE ONLY (see ft8_oracle.c). */
#ifndef FT8_ORACLE_H
#define FT8_ORACLE_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define FT8O_NSAMP      180000
#define FT8O_GRID_ROWS  376      /* row 0 is the never-written 1.0 row; rows 1..375 are hops */
/* compile-time widths, as in include/ft8rx.h: the default build (search_freq_range up to 3000 Hz) and -DFT8O_WIDE (up to 5900 Hz) */
#ifdef FT8O_WIDE
#define FT8O_GRID_COLS  1920
#define FT8O_SPEC_BINS  96000
#define FT8O_MAX_F0     1888
#else
#define FT8O_GRID_COLS  976
#define FT8O_SPEC_BINS  49152    /* cycle-spectrum bins kept (0.0625 Hz each) */
#define FT8O_MAX_F0     960
#endif

/* Mirrors Receiver.__init__ kwargs (reference receiver.py:311-313) + decoder constants
 * (receiver.py:30,78,91,95; decoders.py:223).  Extension knobs are the same fields with
 * other values (BASELINE configs 2/4). */
/* candidates with h0 in this range take the frequency-domain fine sync (the middle Costas block of every tweak lies inside the
 * 3200-sample series); the others the reference's own time-domain form with clamped reads (ft8o_fine) */
#define FT8O_MIN_H0_FD (-140)
#define FT8O_MAX_H0_FD 220

typedef struct {
    float sync_score_min;   /* 85 */
    int32_t max_cands;      /* 200 */
    int32_t f0_lo, f0_hi;   /* 32, 960 : int(100/3.125), int(3000/3.125) */
    int32_t h0_lo, h0_hi;   /* -37, 87 */
    int32_t bp_nc0_a, bp_iters_a;   /* 35, 5  (ipass 0 and 3) */
    int32_t bp_nc0_b, bp_iters_b;   /* 90, 20 (ipass 4) */
    int32_t osd_single, osd_double; /* 30, 2 */
    float llr_sd_min;       /* 5 */
    /* extension knobs with no reference counterpart (BASELINE config 4 "OSD depth-3"); 0 = off = the reference's osd_012 */
    int32_t osd_triple;     /* order-3 reprocessing: triple flips (i, j, k), k < j < i < osd_triple, i-major, tried after the reference's trials */
    int32_t osd_max_hd;     /* acceptance gate: an OSD trial counts (and calls unpack) only if its 174-bit codeword differs from the hard decisions in <= this many positions */
    /* FFT radix plans (0-terminated).  The product exports its plans through the C ABI and the
     * parity tests hand them to the oracle, so both sides run the same butterfly sequence. */
    int32_t plan1920[8], plan3200[8], plan300[8], plan320[8];
} ft8o_config;

/* status codes */
enum { FT8O_ST_NONE = 0, FT8O_ST_DECODED = 1, FT8O_ST_STOP_GRID_SD = 2, FT8O_ST_STOP_COSTAS = 3,
       FT8O_ST_STOP_FINE_SD = 4, FT8O_ST_EXHAUSTED = 5 };
/* method codes */
enum { FT8O_M_GOOD91 = 0, FT8O_M_LDPC_A = 1, FT8O_M_LDPC_B = 2, FT8O_M_OSD = 3, FT8O_M_LDPC_B_OSD = 4 };

typedef struct {
    int32_t f0_idx, h0_idx;
    float score;
    float grid_sd, fine_sd;
    int32_t snr_grid, snr_fine;
    int32_t ttweak, ftweak, nsync;
    int32_t status;         /* FT8O_ST_* */
    int32_t ipass;          /* ipass at which the decode happened (0..6), -1 otherwise */
    int32_t ap;             /* AP pattern index 0..4 */
    int32_t method;         /* FT8O_M_* */
    int32_t n_its;          /* BP iteration index of success, or OSD trial index */
    uint64_t msg_lo, msg_hi;/* 77-bit payload: bit 76 = first transmitted bit */
} ft8o_cand;

/* one CRC-passing unpack() call, in the reference's global call order */
typedef struct {
    uint64_t msg_lo, msg_hi;
    int32_t cand, ipass;
    int32_t valid;          /* unpack returned a tuple */
    int32_t pad;            /* (slot << 16) | seq : attempt index inside the ipass step, order inside the attempt */
} ft8o_event;

typedef struct {
    char f[3][16];          /* msg_tuple */
    int32_t cand;
    int32_t snr;
    double tsec, fHz;       /* after fine update */
    int32_t ipass, ap, method, ttweak, ftweak;
    int32_t fine;           /* source: 0 grid, 1 fine */
} ft8o_msg;

void  ft8o_default_config(ft8o_config* c);
float ft8o_log10f(float x);
float ft8o_tanhf(float x);
void  ft8o_fft(float* data /* interleaved re,im */, int n, const int32_t* plan, float* scratch);
void  ft8o_spectrogram(const int16_t* audio, const ft8o_config* c, float* grid /*[376][976]*/);
int   ft8o_sync_search(const float* grid, const ft8o_config* c, ft8o_cand* out /* >= 960 */);
/* local re-search (receiver_sub.py:434-445): only the columns with mask[f0 - f0_lo] != 0, every score above 0; NULL = configured search */
void  ft8o_set_search_mask(const uint8_t* mask, int32_t n);
void  ft8o_payload(const float* grid, int f0_idx, int h0_idx, float* p /*[58][8]*/);
int   ft8o_db_to_llr(const float* p /*[58][8]*/, float* llr /*[174]*/, float* sd, int32_t* snr);
void  ft8o_cycle_spectrum(const int16_t* audio, const ft8o_config* c, float* spec /*[SPEC_BINS][2]*/);
void  ft8o_fine_grid(const float* spec, const ft8o_config* c, int fb, int tb, float* grid /*[79][8]*/, float* score);
int   ft8o_fine(const float* spec, const ft8o_config* c, int f0_idx, int h0_idx, int32_t* ttweak, int32_t* ftweak,
                int32_t* nsync, float* llr, float* sd, int32_t* snr, float* sgrid /*[79][8] or NULL*/);
void  ft8o_set_ap(const float* llr0, int ap, float* llr);
int   ft8o_crc_valid91(const float* llr91, uint64_t* lo, uint64_t* hi); /* 0 fail, 1 crc ok+invalid, 2 valid */
int   ft8o_valid77(uint64_t lo, uint64_t hi);
int   ft8o_ldpc(float* llr /* in/out */, int max_nc0, int max_iters, uint64_t* lo, uint64_t* hi,
                int32_t* n_its, int32_t* has_out);
int   ft8o_osd_ext(const float* llr, int singles, int doubles, int triples, int max_hd, uint64_t* lo, uint64_t* hi, int32_t* trial,
                   int32_t* info_cols, int32_t* hd_out);
/* np.argsort(x) of n <= 256 float32 values as the reference's numpy (2.2.6, AVX-512: x86-simd-sort) orders them, ties and NaNs included */
int   ft8o_argsort_f32(const float* x, int n, int32_t* out);
int   ft8o_osd(const float* llr, int singles, int doubles, uint64_t* lo, uint64_t* hi, int32_t* trial,
               int32_t* info_cols /*[91] or NULL*/);
/* hash table + rendering (reference decoders.py:16-115, databases.py:8-26) */
void* ft8o_hash_new(void);
void  ft8o_hash_free(void* h);
int   ft8o_unpack77(void* hash, uint64_t lo, uint64_t hi, char out[3][16]);
/* the whole frame: search + ipass ladder in the reference's global order */
int   ft8o_decode_frame(const int16_t* audio, const ft8o_config* c, ft8o_cand* cands, int32_t* n_cands,
                        ft8o_event* log, int32_t log_cap, int32_t* n_log, ft8o_msg* msgs, int32_t msg_cap, int32_t* n_msgs);

/* signal subtraction: Receiver.subtract_signal, reference tests/pipeline/receiver_sub.py:380-402, with
 * symbols_to_complex_audio (PyFT8/transmitter.py:41-70).  audio = the receiver's float32 ring (180000 samples), modified in
 * place; returns 1 if the signal was subtracted, 0 if the reference's guard (sig_s0 > 0, whole signal inside the buffer) fails. */
int   ft8o_subtract(float* audio /*[180000]*/, const uint8_t* tones79, double fHz, double tsec);
/* 77-bit word -> 79 tones (PyFT8/transmitter.py:181-223 encode_bits77) */
void  ft8o_encode_tones(uint64_t lo, uint64_t hi, uint8_t* tones79);
/* AudioIn.get_cycle_spectrum of the subtraction experiment (receiver_sub.py:273-276): the float32 ring, i.e. the residual */
void  ft8o_cycle_spectrum_f32(const float* audio /*[180000]*/, const ft8o_config* c, float* spec /*[SPEC_BINS][2]*/);
/* Candidate.refine_time_origin (receiver_sub.py:58-72) on the float32 residual: updates (fHz, tsec) = ft8rx_subtract refine = 3 */
void  ft8o_refine_time_origin(const float* audio_f32, const ft8o_config* c, double* fHz, double* tsec, float* best_score);
/* the build's own re-estimation on a decimated baseband copy + subtraction (ft8rx_subtract refine = 2; extension): updates
 * (fHz, tsec); subtracts in place if `subtract`; returns 1 if subtracted */
int   ft8o_refine2_subtract(float* audio /*[180000]*/, const uint8_t* tones79, double* fHz, double* tsec, int subtract);

#ifdef __cplusplus
}
#endif
#endif
