/* ft8rx.h -- C ABI of libft8rx.so: the MI355X-native FT8 receive hot path.
 *
 * The reference (G1OJS/PyFT8) is pure Python and has no FFI for this path; its boundary is the
 * Python surface of PyFT8/receiver.py + PyFT8/decoders.py.  Each entry point
 * below names the reference interface it stands in for; pyft8_amd/{receiver,decoders}.py bind
 * them with ctypes and re-create that Python surface.
 *
 * Conventions: plain pointers and sizes, no C++/torch types; every function returns 0 on
 * success and a negative code on failure (ft8rx_last_error() gives the text); nothing throws
 * across the ABI.  One handle == one HIP device + its workspaces; a handle is not thread-safe.
 * Separate handles may be driven from separate threads, but the handles of one process on one
 * device SHARE their HIP streams (a process-wide pool per device: main, second chunk stream, result
 * copy, H2D -- the runtime maps streams onto four hardware queues when they are created, and a
 * later handle's own streams could land two chunk streams on one queue): work of two handles
 * on one device serialises on those streams, ft8rx_sync / ft8rx_destroy of one waits for the
 * other's queued work, and ft8rx_set_profiling stage times include the other handle's kernels.
 * For independent concurrent decoding use one process per GPU (the multi-GPU path does).
 * All device workspaces are allocated at ft8rx_create(); the hot path allocates nothing.
 * Pointers are host pointers unless the parameter is named d_*.
 */
/* MAP OF THE ABI -- which entry points an adopter binds, and which exist for tests and measurements.
 *
 *   CORE (the drop-in boundary, SURVEY.md 8b: everything the Python surface of receiver.py / decoders.py needs)
 *     lifecycle        ft8rx_default_config  ft8rx_create  ft8rx_destroy  ft8rx_last_error  ft8rx_device_count  ft8rx_build_info  ft8rx_build_limits
 *     whole path       ft8rx_decode_batch  ft8rx_decode_messages                       (synchronous: host audio -> records / messages)
 *                      ft8rx_enqueue_batch  ft8rx_enqueue_batch_host  ft8rx_sync  ft8rx_fetch_results  ft8rx_fetch_results_view  (pipelined)
 *     host messages    ft8rx_package_batch  ft8rx_hashes_create / _destroy / _clear / _add / _size   (a14-a17: unpack, call hashes, dict fields)
 *     decoders.py      ft8rx_ldpc  ft8rx_osd  ft8rx_crc_valid  ft8rx_valid77                           (ldpc_decode, osd_012, crc_unpack91, unpack)
 *     streaming        ft8rx_hop_spectrum  ft8rx_sync_search                                       (AudioIn._callback, Receiver.search)
 *     multi-GPU        ft8rx_set_packed_output  ft8rx_packed_results  ft8rx_packed_output_fence  ft8rx_package_packed  ft8rx_alloc_host
 *                      ft8rx_free_host  ft8rx_d2h_async / _query / _event  ft8rx_device_pci_bus_id                                     (the gather of decoded messages)
 *   TUNING AND SERVICE (defaults are the measured best; results never depend on them)
 *     ft8rx_set_streams  ft8rx_set_subbatch  ft8rx_set_ladder_mode  ft8rx_set_reject_log  ft8rx_staging_audio  ft8rx_copy_to_host
 *     ft8rx_results_to_device
 *   EXTENSIONS (no counterpart in the reference's receive path)
 *     ft8rx_synth_frames  ft8rx_synth_frames_ex                                        (f-1: workload generator)
 *     ft8rx_subtract  ft8rx_subtraction_list  ft8rx_encode_tones  ft8rx_merge_messages  ft8rx_set_search_mask   (f-4: subtraction passes)
 *     ft8rx_osd_ext                                                                    (order-3 / distance-gate knobs)
 *   TEST AND MEASUREMENT AIDS (stage entry points of the parity tests, timers, probes -- an adopter never calls these)
 *     ft8rx_spectrogram  ft8rx_sync_scores  ft8rx_llr_grid  ft8rx_cycle_spectrum  ft8rx_fine  ft8rx_get_fft_plans
 *     ft8rx_set_profiling  ft8rx_get_stage_times  ft8rx_math_probe  (and the ft8rx_debug_* symbols of timing-only builds)
 *
 * LIMITS of this build against the reference's open-ended kwargs (pyft8_amd.receiver.config_from_kwargs names the kwarg when one is
 * exceeded; ft8rx_create answers -1): max_cands <= FT8RX_MAX_CANDS = 256 (libft8rx.so) / 2048 (libft8rx_wide.so: more than any search
 * range has f0 bins, i.e. no limit -- a larger max_cands keeps the same candidates; reference: any, default 200), search_time_range inside
 * [-36.4, +22.6] s (FT8RX_MIN_H0 / FT8RX_MAX_H0: beyond it the reference's own search raises IndexError), search_freq_range 12.5 .. 3000 Hz (libft8rx.so) / .. 5900 Hz
 * (libft8rx_wide.so), OSD flip counts <= 91 and <= 16384 trials.
 */
#ifndef FT8RX_H
#define FT8RX_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define FT8RX_NSAMP      180000   /* 15 s @ 12 kHz, int16 mono (receiver.py:9-12,241) */
#define FT8RX_GRID_ROWS  376      /* row 0 = never-written 1.0 row, rows 1..375 = hops (receiver.py:237-240) */
/* The spectrogram and cycle-spectrum layouts are compile-time widths.  The library is built twice from the same source:
 *   libft8rx.so       search_freq_range up to 3000 Hz, the reference's default (receiver.py:312): f0_hi <= 960
 *   libft8rx_wide.so  (-DFT8RX_WIDE) search_freq_range up to 5900 Hz: f0_hi <= 1888 -- twice the grid / spectrum memory per frame --
 *                     and up to 2048 candidates per frame (every f0 bin of any search range can pass sync_score_min: 928 at the
 *                     default range, receiver.py:338-367), eight times the per-candidate workspaces
 * Same ABI; ft8rx_build_info() tells which one is loaded.  (The reference allocates f0_hi + 16 grid columns, receiver.py:240, and
 * fails beyond ~5940 Hz, where the fine-sync slice runs off the 96001-bin cycle spectrum, receiver.py:181-182.) */
#ifdef FT8RX_WIDE
#define FT8RX_GRID_COLS  1920     /* bins 0..1919 of the 3840-point real FFT */
#define FT8RX_SPEC_BINS  96000    /* bins 0..95999 of the 192000-point real FFT */
#define FT8RX_MAX_F0     1888
#define FT8RX_MAX_CANDS  2048     /* upper bound for config.max_cands = the per-frame stride of the per-candidate workspaces */
#else
#define FT8RX_GRID_COLS  976      /* receiver.py:240 at the default range: 960 + 16 */
#define FT8RX_SPEC_BINS  49152    /* kept bins of the 192000-point cycle spectrum (receiver.py:280-286) */
#define FT8RX_MAX_F0     960
#define FT8RX_MAX_CANDS  256      /* upper bound for config.max_cands (the reference's default is 200, receiver.py:311) */
#endif
#define FT8RX_MIN_H0     (-898)   /* bounds of config.h0_lo / h0_hi (search_time_range -36.4 .. +22.6 s; the reference's default is -2 .. +3 s): where the   */
#define FT8RX_MAX_H0     578      /* reference's own search stops indexing its 750-row grid (rows h0 + 148 .. h0 + 172, receiver.py:322, 346-347)            */
#define FT8RX_MIN_H0_FD  (-140)   /* candidates with h0 in [MIN_H0_FD, MAX_H0_FD] (-6.1 .. +8.3 s): the middle Costas block of every tweak lies inside the      */
#define FT8RX_MAX_H0_FD  220      /* 3200-sample fine-sync series and the frequency-domain scores / grid apply; further out the reads are clamped (:189-195)   */
                                  /* and the candidate is scored in the time domain, one series per tweak (kernels/fine_sync.hpp: k_fine_td)                 */
#define FT8RX_EVENT_CAP  512      /* per-frame capacity of the CRC-pass event log */

/* Receiver(...) kwargs (receiver.py:311-313) + module/decoder constants (receiver.py:30,78,91,95;
 * decoders.py:223).  BASELINE "30 iters / OSD depth" extension knobs are these same fields. */
typedef struct {
    float   sync_score_min;          /* 85 */
    int32_t max_cands;               /* 200 */
    int32_t f0_lo, f0_hi;            /* 32, 960 = int(search_freq_range / 3.125) */
    int32_t h0_lo, h0_hi;            /* -37, 87 = int((search_time_range + 0.5) * 25) */
    int32_t bp_nc0_a, bp_iters_a;    /* 35, 5  : ldpc_decode(llr, 35, 5)  (ipass 0 and 3) */
    int32_t bp_nc0_b, bp_iters_b;    /* 90, 20 : ldpc_decode(llr, 90, 20) (ipass 4) */
    int32_t osd_single, osd_double;  /* 30, 2  : osd_012(llr, 30, 2) */
    float   llr_sd_min;              /* 5 : Candidate.llr_sd_min */
    /* Extension knobs with no reference counterpart (BASELINE config 4 "OSD depth-3"); 0 = off = the reference's osd_012. */
    int32_t osd_triple;              /* 0 : order-3 reprocessing -- after the reference's trials, triple flips (i, j, k), k < j < i < osd_triple
                                      *     over the least reliable basis positions, i-major; C(osd_triple, 3) extra trials (<= 40) */
    int32_t osd_max_hd;              /* 0 : acceptance gate -- an OSD trial counts (and calls unpack) only if its 174-bit codeword differs from
                                      *     the hard decisions in <= osd_max_hd positions.  The reference accepts the first CRC-valid trial
                                      *     whatever its distance (decoders.py:248-272), which is where its false decodes come from. */
} ft8rx_config;

enum { FT8RX_ST_ACTIVE = 0, FT8RX_ST_DECODED = 1, FT8RX_ST_STOP_GRID_SD = 2, FT8RX_ST_STOP_COSTAS = 3,
       FT8RX_ST_STOP_FINE_SD = 4, FT8RX_ST_EXHAUSTED = 5 };
enum { FT8RX_M_GOOD91 = 0, FT8RX_M_LDPC_A = 1, FT8RX_M_LDPC_B = 2, FT8RX_M_OSD = 3, FT8RX_M_LDPC_B_OSD = 4 };

/* One candidate's outcome (replaces the state of a reference `Candidate`, receiver.py:29-66). 48 bytes. */
typedef struct {
    uint64_t msg_lo, msg_hi;         /* 77-bit payload, bit 76 = first transmitted bit; valid iff status==DECODED */
    float    score, grid_sd, fine_sd;
    int16_t  f0_idx, h0_idx;
    int8_t   ttweak, ftweak, snr_grid, snr_fine;
    uint8_t  status, ipass, ap, method;
    int16_t  n_its;                  /* BP iteration index or OSD trial index of the success */
    uint8_t  nsync;
    uint8_t  osd_hd;                 /* OSD decodes: Hamming distance of the accepted codeword to the 174 hard decisions */
    uint32_t pad2;
} ft8rx_record;

/* One CRC-14-passing, non-zero 77-bit word met on the decode ladder (= one call of the reference's
 * unpack(), decoders.py:131).  The host replays these in reference order to reproduce the call-hash
 * table side effects (databases.py:10-26).  24 bytes. */
typedef struct {
    uint64_t msg_lo, msg_hi;
    uint16_t cand;                   /* candidate index within the frame */
    uint8_t  ipass;                  /* ladder step the call belongs to (0..6) */
    uint8_t  slot;                   /* attempt index inside the ipass step (AP index / saved-llr index) */
    uint16_t seq;                    /* order inside the attempt: GOOD91=0, BP iteration+1, OSD trial */
    uint16_t valid;                  /* unpack() would have returned a tuple */
} ft8rx_event;

/* One emitted message (the argument of the reference's on_message callback, receiver.py:61-65, before string
 * formatting).  64 bytes. */
typedef struct {
    char     f[3][16];               /* msg_tuple */
    int16_t  cand, f0_idx, h0_idx;
    int8_t   snr, ttweak, ftweak;
    uint8_t  ipass, ap, method, fine;
    uint8_t  pad[3];
} ft8rx_message;

typedef struct ft8rx_handle ft8rx_handle;
typedef struct ft8rx_hashes ft8rx_hashes;                           /* persistent call-hash table, ft8rx_hashes_* below */

/* ---- lifecycle -------------------------------------------------------------------------- */
int  ft8rx_default_config(ft8rx_config* cfg);                       /* Receiver.__init__ defaults */
int  ft8rx_create(const ft8rx_config* cfg, int device, int max_frames, ft8rx_handle** out);
void ft8rx_destroy(ft8rx_handle* h);
const char* ft8rx_last_error(ft8rx_handle* h);                      /* h may be NULL: last create error */
int  ft8rx_device_count(void);
/* PCI address of a device ("0000:c1:00.0"; buf >= 16 bytes): lets a multi-rank host place each rank's threads and page-locked
 * buffers on the GPU's NUMA node (/sys/bus/pci/devices/<address>/numa_node), as bench.py does */
int  ft8rx_device_pci_bus_id(int device, char* buf, int len);
/* the compile-time widths of the loaded library (FT8RX_GRID_COLS, FT8RX_SPEC_BINS, FT8RX_MAX_F0): 976 / 49152 / 960, or
 * 1920 / 96000 / 1888 for the wide build */
int  ft8rx_build_info(int32_t* grid_cols, int32_t* spec_bins, int32_t* max_f0);
/* ... and its capacities (FT8RX_MAX_CANDS, FT8RX_EVENT_CAP): 256 / 512, or 2048 / 512 for the wide build */
int  ft8rx_build_limits(int32_t* max_cands, int32_t* event_cap);
/* FFT radix plans the kernels use (0-terminated, <= 8 entries each): 1920, 3200, 300, 320 point */
int  ft8rx_get_fft_plans(int32_t* p1920, int32_t* p3200, int32_t* p300, int32_t* p320);

/* ---- whole hot path: audio -> candidate records ------------------------------------------
 * Stands in for AudioIn._callback x375 + Receiver.search + the Candidate.decode ladder driven by
 * Receiver.manage_cycle (receiver.py:295-306, 338-367, 68-107, 389-398) under frame-complete
 * semantics, for n_frames independent 15-s frames.
 *   records : [n_frames][cfg.max_cands]   counts : [n_frames]
 *   events  : [n_frames][FT8RX_EVENT_CAP] event_counts : [n_frames] (may exceed the cap => truncated) */
int  ft8rx_decode_batch(ft8rx_handle* h, const int16_t* audio, int n_frames,
                        ft8rx_record* records, int32_t* counts, ft8rx_event* events, int32_t* event_counts);
/* Same, split for pipelining: audio already resident in HBM (device pointer); enqueue is asynchronous.  Results are double
 * buffered: when a batch's kernels finish, a copy stream moves its records/events into page-locked host buffers while the next
 * enqueued batch computes.  ft8rx_fetch_results waits for and returns the OLDEST unfetched batch (or the latest batch again if
 * all have been fetched); at most two batches are retained -- a third enqueue drops the oldest.  Steady state:
 *   enqueue(0); for k = 1..: enqueue(k); fetch(k-1); <host message layer of k-1>   -- the GPU never waits for the host. */
int  ft8rx_enqueue_batch(ft8rx_handle* h, const int16_t* d_audio, int n_frames);
/* Ordering: consecutive enqueued batches of the same size run as free-running chunk streams (chunk i of batch k+1 follows chunk i of
 * batch k on its own HIP stream, no per-batch fork / join; only the result copy waits for all chunks).  Every other entry point of
 * the handle -- stage functions, ft8rx_decode_batch, ft8rx_subtract, profiling passes -- first waits until the batches in flight are
 * complete, so mixing them with enqueue / fetch is safe (and costs that wait). */
/* The same pipeline fed from HOST memory: the audio of batch k+1 is copied (in chunks, on a dedicated stream, into the second of two
 * device staging buffers) while batch k computes, so in steady state  enqueue_host(k+1); fetch(k); <host layer of k>  hides the PCIe
 * transfer behind the kernels.  `audio` must stay valid until the batch has been fetched; page-locked memory (ft8rx_alloc_host) makes
 * the copies truly asynchronous. */
int  ft8rx_enqueue_batch_host(ft8rx_handle* h, const int16_t* audio, int n_frames);
int  ft8rx_sync(ft8rx_handle* h);
int  ft8rx_fetch_results(ft8rx_handle* h, int n_frames, ft8rx_record* records, int32_t* counts,
                         ft8rx_event* events, int32_t* event_counts);
/* Only events[f][0 .. min(event_counts[f], FT8RX_EVENT_CAP)) are written / valid -- here, in the view below and in
 * ft8rx_decode_batch; the rest of a frame's row is left as it was.  For batches whose event log exceeds 1 MB the GPU packs the used
 * entries (the log is 12 KB per frame, a tenth of it used) straight into page-locked host memory: eight ranks sharing the host
 * links move ~8 MB instead of 100 MB per 8192-frame shard. */
/* Zero-copy variant of ft8rx_fetch_results: waits for the same batch and returns pointers INTO the handle's page-locked result
 * buffers (records packed [n_frames][cfg.max_cands], events [n_frames][FT8RX_EVENT_CAP]).  They stay valid until two more
 * batches have been enqueued (the slot is then reused). */
int  ft8rx_fetch_results_view(ft8rx_handle* h, int n_frames, const ft8rx_record** records, const int32_t** counts,
                              const ft8rx_event** events, const int32_t** event_counts);
/* Multi-GPU gather support: copies the LATEST batch's results, device to device, into caller-owned device buffers laid out like
 * the host outputs of ft8rx_decode_batch (records packed [n_frames][cfg.max_cands]); waits for the batch first.  The caller hands
 * these buffers to its collective (RCCL gather over xGMI, pyft8_amd/distributed.py) -- no host round trip.  Any pointer may be NULL. */
int  ft8rx_results_to_device(ft8rx_handle* h, int n_frames, ft8rx_record* d_records, int32_t* d_counts,
                             ft8rx_event* d_events, int32_t* d_event_counts);
/* ---- packed results: what a multi-GPU gather moves (the reference has no counterpart) -------------------------
 * The dense result arrays are [n_frames][max_cands] records + [n_frames][512] events = up to 24 KB per frame, of which the host
 * message layer reads ~4 KB: the records of the candidates that DECODED or made at least one unpack() call (an event), and the
 * used part of the event log.  With a packed output set, every enqueued batch ends with three small kernels that write exactly
 * that, back to back, into the caller's buffer of the batch's result slot:
 *     ft8rx_packed_header | ft8rx_packed_frame[n_frames] | ft8rx_record[n_records] | ft8rx_event[n_events]
 * frame f's records are records[rec_off .. rec_off + n_rec) in candidate order, each with pad2 = its candidate index inside the
 * frame; its events are events[ev_off .. ev_off + min(n_ev, FT8RX_EVENT_CAP)).  ft8rx_package_packed renders the same messages from
 * this as ft8rx_package_batch does from the dense arrays (the replay only ever looks at those candidates), so a rank sends ~4 KB
 * per frame to rank 0 instead of 13-24 KB and rank 0 keeps the packed form (pyft8_amd/distributed.py: PackedGather). */
#define FT8RX_PACKED_MAGIC 0x50385446u      /* "FT8P" */
typedef struct {
    uint32_t magic;                  /* FT8RX_PACKED_MAGIC */
    int32_t  n_frames;
    int32_t  n_records, n_events;    /* totals over the batch */
    uint64_t bytes;                  /* header + frame table + records + events */
    int32_t  max_cands;              /* cfg.max_cands of the producing handle */
    int32_t  overflow;               /* != 0: `bytes` exceeds the buffer's capacity -- only header and frame table were written */
} ft8rx_packed_header;               /* 32 bytes */
typedef struct {
    int32_t  rec_off, ev_off;        /* first record / event of the frame in the packed runs */
    uint16_t n_cand;                 /* the frame's candidate count (counts[f] of the dense form) */
    uint16_t n_rec;                  /* records kept: decoded or with at least one event */
    int32_t  n_ev;                   /* event_counts[f] of the dense form (may exceed FT8RX_EVENT_CAP: the log overflowed) */
} ft8rx_packed_frame;                /* 16 bytes */
/* d_buf0 / d_buf1: one buffer per result slot (batches alternate), cap_bytes each, device memory or page-locked host memory
 * (ft8rx_alloc_host: the kernels then write straight across PCIe); NULL, NULL turns the packed output off.  Worst case per frame:
 * 16 + 48 max_cands + 24 x 512 bytes; config 1 frames need ~4 KB.  Applies to batches enqueued afterwards. */
int  ft8rx_set_packed_output(ft8rx_handle* h, void* d_buf0, void* d_buf1, uint64_t cap_bytes);
/* The packed output of the batch the last ft8rx_fetch_results / _view / ft8rx_decode_batch call returned (that call has waited for
 * it): which of the two buffers (0 / 1) and a copy of its header.  -1 if no packed output was set when that batch was enqueued. */
int  ft8rx_packed_results(ft8rx_handle* h, int32_t* which, ft8rx_packed_header* header);
/* A consumer that reads packed buffer `which` (0 / 1) ASYNCHRONOUSLY on a stream of its own (an RCCL send) hands over a HIP event
 * (hipEvent_t) recorded on that stream behind its last read: the next batch that packs into this buffer -- two enqueues later --
 * makes its pack kernels wait for the event on the device.  One pending fence per buffer; the event must stay alive until that
 * batch has been enqueued.  NULL clears it. */
int  ft8rx_packed_output_fence(ft8rx_handle* h, int which, void* hip_event);
/* Host side (no GPU needed): messages of frames [frame_lo, frame_lo + n_frames) of a packed buffer, as ft8rx_package_batch renders
 * them from the dense arrays (n_threads / table / flags as there); out [n_frames][max_msgs].  Returns -1 for a malformed or
 * overflowed buffer. */
int  ft8rx_package_packed(const void* packed, uint64_t bytes, int frame_lo, int n_frames, ft8rx_message* out, int max_msgs,
                          int32_t* out_counts, int n_threads, ft8rx_hashes* table, int32_t* flags);
/* per-kernel HIP-event timing of the most recent enqueue (enable before enqueue). names/ms: up to 16 */
/* number of HIP streams a batch is cut across (1..8, default 2: measured best, profiles/archive/r02_notes.md); profiling mode always uses one */
int  ft8rx_set_streams(ft8rx_handle* h, int n);
/* frames per kernel chain inside a stream's share of a batch (default FT8RX_SUBBATCH_DEFAULT; 0 = the whole share in one chain): a
 * large batch runs as a sequence of cache-sized sub-batches, each through the whole path before the next starts, so that a stage
 * reads what the stage before it wrote from L2 / MALL instead of HBM.  Records, events and messages do not depend on it. */
#define FT8RX_SUBBATCH_DEFAULT 256    /* measured: 128 .. 256 are equal on dense frames (BASELINE configs 2, 3), 256 is 10 % better on sparse ones (config 4) */
int  ft8rx_set_subbatch(ft8rx_handle* h, int frames);
/* how the fine-stage BP attempts of a batch are launched; records and messages are identical either way:
 * 0 (default) = in the reference's ladder order (receiver.py:84-98) as three launches, candidates that are decided dropping out in
 *     between -- least work, highest throughput;
 * 1 = all five AP variants in one launch -- one dependent BP instead of three: lower latency for batches too small to fill the GPU
 *     (one frame: 0.38 vs 0.49 ms host to host).  The event log then also holds CRC-passing words of attempts the ladder would not
 *     have reached; ft8rx_package_batch skips them. */
int  ft8rx_set_ladder_mode(ft8rx_handle* h, int mode);
/* Local re-search of the reference's subtraction experiment (tests/pipeline/receiver_sub.py:434-445: after a signal has been
 * subtracted, search(f0_idx - 2 .. f0_idx + 1, ignore_sync_score_min = True)): mask[n_frames][cfg.f0_hi - cfg.f0_lo], one byte per
 * search column.  While a mask is set, the candidate selection of every batch (Receiver.search, receiver.py:338-367) takes ONLY the
 * columns whose byte is non-zero and every score above 0 instead of above sync_score_min; order (score descending, stable) and the
 * max_cands cap as always.  mask = NULL: back to the configured search.  Batches in flight finish under the setting they started with. */
int  ft8rx_set_search_mask(ft8rx_handle* h, const uint8_t* mask, int n_frames);
int  ft8rx_set_profiling(ft8rx_handle* h, int on);
int  ft8rx_get_stage_times(ft8rx_handle* h, int* n, const char** names, float* ms);

/* ---- stage entry points (parity tests; each mirrors one reference function) ---------------- */
/* AudioIn.get_hop_spectrum x375 (receiver.py:288-293): grid [n][376][FT8RX_GRID_COLS] */
int  ft8rx_spectrogram(ft8rx_handle* h, const int16_t* audio, int n_frames, float* grid);
/* streaming mode: ONE call of AudioIn.get_hop_spectrum (receiver.py:288-293): the last 3840 int16 samples -> FT8RX_GRID_COLS dB values */
int  ft8rx_hop_spectrum(ft8rx_handle* h, const int16_t* window3840, float* row);
/* Receiver.search (receiver.py:338-367): per frame <= max_cands (f0,h0,score), sorted */
int  ft8rx_sync_search(ft8rx_handle* h, const float* grid, int n_frames,
                       int32_t* f0_idx, int32_t* h0_idx, float* score, int32_t* counts);
/* The inner loops of Receiver.search (receiver.py:341-349) for ANY f0 index range [f0_lo, f0_hi) the grid can hold (4 <= f0_lo,
 * f0_hi <= FT8RX_GRID_COLS - 15), whatever range the handle was created for: per frame and f0 the first strict maximum of the Costas
 * score over the handle's h0 range, starting from 0 -- score [n][f0_hi - f0_lo] and its h0 (0 / 0 where no score is positive).  The
 * threshold, stable sort and cut (receiver.py:350-367) are the caller's; Receiver.search uses this for `search_f_idxs` lists that
 * are not the configured range. */
int  ft8rx_sync_scores(ft8rx_handle* h, const float* grid, int n_frames, int f0_lo, int f0_hi, float* score, int32_t* h0_idx);
/* Candidate._get_llr_grid/_dB_to_llr (receiver.py:136-138, 208-222) for n (frame,f0,h0) triples */
int  ft8rx_llr_grid(ft8rx_handle* h, const float* grid, int n_frames, int n, const int32_t* frame,
                    const int32_t* f0_idx, const int32_t* h0_idx, float* llr /*[n][174]*/, float* sd, int32_t* snr);
/* AudioIn.get_cycle_spectrum (receiver.py:280-286): spec [n][FT8RX_SPEC_BINS] complex64 */
int  ft8rx_cycle_spectrum(ft8rx_handle* h, const int16_t* audio, int n_frames, float* spec);
/* Candidate._get_llr_fine (receiver.py:140-206) for n (frame,f0,h0) triples; sgrid [n][79][8] may be NULL.
 * ret[i]: 1 continue, 0 Costas gate failed, -1 sd gate failed */
int  ft8rx_fine(ft8rx_handle* h, const float* spec, int n_frames, int n, const int32_t* frame,
                const int32_t* f0_idx, const int32_t* h0_idx, int32_t* ret, int32_t* ttweak, int32_t* ftweak,
                int32_t* nsync, float* llr, float* sd, int32_t* snr, float* sgrid);
/* ldpc_decode(llr, max_ncheck0, max_iters) (decoders.py:153-171) on n vectors.
 * ok[i]=1 => msg; has_out[i]=1 => llr_out[i] holds the mutated llr (the reference's third return) */
int  ft8rx_ldpc(ft8rx_handle* h, const float* llr, int n, int max_ncheck0, int max_iters,
                int32_t* ok, uint64_t* msg_lo, uint64_t* msg_hi, int32_t* n_its, int32_t* has_out, float* llr_out);
/* osd_012(llr, singleflips, doubleflips) (decoders.py:223-272) on n vectors; 0 <= singleflips, doubleflips <= 91 (all basis positions) */
int  ft8rx_osd(ft8rx_handle* h, const float* llr, int n, int singleflips, int doubleflips,
               int32_t* ok, uint64_t* msg_lo, uint64_t* msg_hi, int32_t* trial);
/* same with the build's extension knobs: tripleflips = order-3 depth, max_hd = acceptance gate (0 = off); hd[i] (optional) = the
 * Hamming distance of the accepted codeword to the hard decisions */
int  ft8rx_osd_ext(ft8rx_handle* h, const float* llr, int n, int singleflips, int doubleflips, int tripleflips, int max_hd,
                   int32_t* ok, uint64_t* msg_lo, uint64_t* msg_hi, int32_t* trial, int32_t* hd);
/* crc_unpack91 (decoders.py:117-131) on n x 91 soft/hard values: res 0 = no CRC, 1 = CRC ok but unpack None, 2 = tuple */
int  ft8rx_crc_valid(ft8rx_handle* h, const float* cw91, int n, int32_t* res, uint64_t* msg_lo, uint64_t* msg_hi);
/* unpack() validity predicate (decoders.py:16-115) on n 77-bit words */
int  ft8rx_valid77(ft8rx_handle* h, const uint64_t* msg_lo, const uint64_t* msg_hi, int n, int32_t* valid);
/* Workload generator (modelled on transmitter.py:41-70): n_frames synthetic 15-s frames of
 * n_signals GFSK signals + Philox white noise, written to the device buffer d_audio[n_frames][180000] int16.
 * signal_table: n_frames*n_signals records laid out as pyft8_amd/synth.py:SIGNAL_DTYPE; pulse_cumsum: 5761 doubles. */
int  ft8rx_synth_frames(ft8rx_handle* h, uint64_t seed, int first_index, int n_frames, int n_signals,
                        const void* signal_table, int signal_bytes, const double* pulse_cumsum, int16_t* d_audio);
/* same; no_noise != 0 leaves the Philox noise out (parity tests of the signal part against the numpy twin pyft8_amd/synth.py) */
int  ft8rx_synth_frames_ex(ft8rx_handle* h, uint64_t seed, int first_index, int n_frames, int n_signals,
                           const void* signal_table, int signal_bytes, const double* pulse_cumsum, int16_t* d_audio, int no_noise);
/* Host message layer (pure host code, no GPU needed): replays each frame's records + events in the reference's emit
 * order -- hash-table side effects of every unpack() call (decoders.py:44,92; databases.py:10-26), duplicate filter
 * (receiver.py:51-66), round/llr_sd ordering of manage_cycle (receiver.py:389-398) -- and renders the message tuples.
 * records [n_frames][max_cands], events [n_frames][FT8RX_EVENT_CAP]; out [n_frames][max_msgs]; out_counts[f] <= max_msgs.
 *   table == NULL : every frame gets a fresh call-hash table (independent batched frames, the parity semantics of
 *                   DESIGN.md section 1); frames are spread over n_threads host threads.
 *   table != NULL : frames are replayed in order on the caller's thread against that persistent table, which they update --
 *                   the reference's process-global `call_hashes` (databases.py:8): a hashed / non-standard call heard in
 *                   cycle N resolves `<...>` in cycle N+1.  This is what the streaming receiver uses.
 * flags (optional, [n_frames]): FT8RX_PKG_* bits per frame.
 * The packager runs its frames on a persistent pool of worker threads owned by the library; ONE ft8rx_package_batch /
 * ft8rx_package_packed call uses the pool at a time, concurrent callers of a process queue behind each other. */
#define FT8RX_PKG_MSG_TRUNCATED    1   /* more than max_msgs messages: the list was cut (size max_msgs >= cfg.max_cands to rule it out) */
#define FT8RX_PKG_EVENTS_TRUNCATED 2   /* event_counts[f] > FT8RX_EVENT_CAP: unpack() calls were dropped, `<...>` strings may differ */
int  ft8rx_package_batch(const ft8rx_record* records, const int32_t* counts, const ft8rx_event* events, const int32_t* event_counts,
                         int n_frames, int max_cands, ft8rx_message* out, int max_msgs, int32_t* out_counts, int n_threads,
                         ft8rx_hashes* table, int32_t* flags);
/* Multi-pass decoding (extension; no GPU needed): append to each frame's message list out[f][0..out_counts[f])
 * those messages of a later pass, add[f][0..add_counts[f]), whose text the frame does not have yet (pad[0] = pass_tag marks them);
 * the same messages, untagged, go to fresh[f] (optional; [n_frames][max_add]) -- the input of the next subtraction sweep.
 * drop_osd != 0 ignores the later pass's OSD decodes (first CRC-valid trial wins there: the source of false decodes). */
int  ft8rx_merge_messages(ft8rx_message* out, int32_t* out_counts, int max_out, const ft8rx_message* add, const int32_t* add_counts,
                          int max_add, int n_frames, int pass_tag, int drop_osd, ft8rx_message* fresh, int32_t* fresh_counts);
/* The whole drop-in in one call -- what a compiled-language binding (cgo / JNI / N-API) would bind: host audio [n_frames][180000]
 * int16 in, the frames' messages out (Receiver's on_message payloads before string formatting, in the reference's emit order):
 * ft8rx_decode_batch followed by ft8rx_package_batch on the handle's own result buffers.  out: [n_frames][max_msgs] (max_msgs >=
 * cfg.max_cands rules truncation out); n_threads / table / flags as for ft8rx_package_batch. */
int  ft8rx_decode_messages(ft8rx_handle* h, const int16_t* audio, int n_frames, ft8rx_message* out, int max_msgs, int32_t* out_counts,
                           int n_threads, ft8rx_hashes* table, int32_t* flags);
/* the persistent call-hash table (databases.py:8-26 `call_hashes` + add_call_hashes) */
ft8rx_hashes* ft8rx_hashes_create(void);
void ft8rx_hashes_destroy(ft8rx_hashes* t);
int  ft8rx_hashes_clear(ft8rx_hashes* t);
int  ft8rx_hashes_add(ft8rx_hashes* t, const char* call);           /* add_call_hashes(call) */
int  ft8rx_hashes_size(const ft8rx_hashes* t);                       /* number of (hash, nbits) keys */
/* Optional reject log: with a path set, every callsign that fails the plausibility test is appended to that file, one per line,
 * as the reference does unconditionally to ./rejected_callsigns.txt (decoders.py:114-115).  NULL or "" turns it off (default). */
int  ft8rx_set_reject_log(const char* path);
/* the handle's own device audio buffer ([max_frames][180000] int16) and a D2H copy helper (tests, tools) */
int16_t* ft8rx_staging_audio(ft8rx_handle* h);
int  ft8rx_copy_to_host(ft8rx_handle* h, void* dst, const void* d_src, uint64_t bytes);
/* Asynchronous device -> page-locked-host copies on the handle's result-copy stream (the one stream no decode kernel waits behind; a
 * copy on any other stream of the process shares a hardware queue with a decode stream and holds its kernels back while it runs).
 * For consumers of device-side results -- the gather on rank `dst` (pyft8_amd/distributed.py).  32 tickets: the 33rd call waits for the oldest.
 *   ft8rx_d2h_async: enqueue; *ticket identifies the copy     ft8rx_d2h_query: 1 landed, 0 not yet, < 0 error
 *   ft8rx_d2h_event: the hipEvent_t recorded behind the copy (valid until 32 more copies were issued), e.g. for ft8rx_packed_output_fence */
int   ft8rx_d2h_async(ft8rx_handle* h, void* dst, const void* d_src, uint64_t bytes, int32_t* ticket);
int   ft8rx_d2h_query(ft8rx_handle* h, int32_t ticket);
void* ft8rx_d2h_event(ft8rx_handle* h, int32_t ticket);
/* ---- signal subtraction ------------------------------------------------------
 * One decoded signal to remove: its 79 tones (Costas + Gray-coded codeword), refined frequency and start time.  96 bytes. */
typedef struct {
    double  fHz, tsec;               /* candidate origin after fine sync (receiver.py:166) */
    uint8_t tones[79];
    uint8_t pad;
} ft8rx_subsig;
/* Stands in for Receiver.subtract_signal of the reference's subtraction experiment (tests/pipeline/receiver_sub.py:380-402,
 * with PyFT8/transmitter.py:41-70 for the GFSK model): for every frame, signals sigs[frame][0 .. counts[frame]) are subtracted in
 * list order from a float32 working copy of the device-resident int16 audio; the residual is rounded back to int16 in place
 * (d_audio is a device pointer).  audio_f32_out (host, optional, [n_frames][180000]) receives the float32 residual. */
int  ft8rx_subtract(ft8rx_handle* h, int16_t* d_audio, int n_frames, ft8rx_subsig* sigs, const int32_t* counts,
                    int max_sigs, int refine, float* audio_f32_out);
/* refine = 0: the given (fHz, tsec) are used as they are (the reference's arithmetic).  refine = 1 (extension): before a signal is
 * subtracted its origin is re-estimated with the same signal model over all 79 symbols -- start sample within [-150, +20] ms and
 * frequency within [-1.75, +5.75] Hz of the given values -- and `sigs` is updated with the refined origins.  (The decoder's
 * tsec/fHz follow the search grid's conventions and sit ~75 ms / ~1.9 Hz off the true start; cancellation needs a few ms.)
 * refine = 2 (extension): the same re-estimation on a copy of the residual that is mixed down to the signal's centre frequency and
 * decimated by 32 (time grid 2.67 ms): same accuracy and decode yield, a third of the time; the subtraction itself stays at full
 * rate with the exact model.  This is what Receiver's multi-pass decode uses.
 * refine = 3: Candidate.refine_time_origin of the reference's experiment (receiver_sub.py:58-72) -- before each signal is subtracted
 * its start time is re-estimated in 12 steps of 5 ms from tb_0 - 6 on the spectrum of the residual so far (untapered 1000-bin slice,
 * score = max over the three Costas blocks), tsec = tb / 200, fHz = int(0.5 + 16 fHz) / 16 -- then subtract_signal as with refine = 0.
 * `sigs` returns the re-estimated origins. */
/* Multi-pass decoding: the signals a subtraction sweep removes = every message of a frame with snr > min_snr, in emit order, with the
 * tones of its codeword (ft8rx_encode_tones of the candidate's word) and the origin its message dict reports (receiver.py:166).
 * sigs: [n_frames][max_sigs]; returns the largest per-frame count (or < 0). */
int  ft8rx_subtraction_list(const ft8rx_message* msgs, const int32_t* counts, int max_msgs, const ft8rx_record* records, int max_cands,
                            int n_frames, int min_snr, ft8rx_subsig* sigs, int max_sigs, int32_t* sig_counts);
/* 77-bit words -> the 79 transmitted tones (CRC-14, LDPC(174,91) encode, Gray map, Costas framing; reference
 * transmitter.py:181-223 `encode_bits77`).  Host function, no GPU.  tones: [n][79]. */
int  ft8rx_encode_tones(const uint64_t* msg_lo, const uint64_t* msg_hi, int n, uint8_t* tones);
/* Page-locked host memory for audio handed to ft8rx_decode_batch: copies from it are true asynchronous DMA that overlaps the
 * kernels of earlier chunks (pageable memory works too, at a lower PCIe-inclusive rate).  Free with ft8rx_free_host. */
void* ft8rx_alloc_host(ft8rx_handle* h, uint64_t bytes);
int  ft8rx_free_host(ft8rx_handle* h, void* p);               /* h may be NULL */
/* arithmetic-contract probes: which 0 = log10f, 1 = tanhf; 2 = forward FFT of length n (x = interleaved complex) */
int  ft8rx_math_probe(ft8rx_handle* h, int which, const float* x, int n, float* y);

#ifdef __cplusplus
}
#endif
#endif
