/* asan_driver.c -- host sanitizer target for the CPU oracle (-fsanitize=address,undefined for host C/C++).
 * TEST INFRASTRUCTURE.  Decodes the 16-bit mono WAV files (or raw int16 frames) named on the command line, plus one frame of
 * digital silence, with the reference knobs and with the extension knobs (order-3 OSD + distance gate), under ASan + UBSan.
 *   make -C oracle asan && oracle/_build/asan_oracle tests/golden/test_08.wav */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "ft8_oracle.h"

static int load(const char* path, int16_t* out) {
    FILE* f = fopen(path, "rb");
    if (!f) return -1;
    unsigned char hdr[44];
    size_t skip = 0;
    if (fread(hdr, 1, 44, f) == 44 && !memcmp(hdr, "RIFF", 4) && !memcmp(hdr + 8, "WAVE", 4)) skip = 44;
    fseek(f, (long)skip, SEEK_SET);
    memset(out, 0, sizeof(int16_t) * FT8O_NSAMP);
    size_t n = fread(out, sizeof(int16_t), FT8O_NSAMP, f);
    fclose(f);
    return (int)n;
}

static ft8o_cand cands[256]; static ft8o_event evlog[4096]; static ft8o_msg msgs[256];
static int run(const int16_t* audio, const ft8o_config* cfg) {
    int32_t nc = 0, nl = 0, nm = 0;
    ft8o_decode_frame(audio, cfg, cands, &nc, evlog, 4096, &nl, msgs, 256, &nm);
    return nm;
}
/* the subtraction functions on the first decodes of the frame: tone encoder, the experiment's refine_time_origin, the
 * build's decimated-baseband re-estimation + subtraction, the reference-style subtraction; incl. origins at the edges of the buffer */
static int run_sub(const int16_t* audio, const ft8o_config* cfg) {
    static float wf[FT8O_NSAMP];
    int32_t nc = 0, nl = 0, nm = 0, done = 0;
    ft8o_decode_frame(audio, cfg, cands, &nc, evlog, 4096, &nl, msgs, 256, &nm);
    for (int i = 0; i < FT8O_NSAMP; i++) wf[i] = (float)audio[i];
    for (int i = 0; i < nm && i < 3; i++) {
        uint8_t tones[79];
        const ft8o_cand* c = &cands[msgs[i].cand];
        ft8o_encode_tones(c->msg_lo, c->msg_hi, tones);
        double f = msgs[i].fHz, t = msgs[i].tsec; float sc;
        ft8o_refine_time_origin(wf, cfg, &f, &t, &sc);
        f = msgs[i].fHz; t = msgs[i].tsec;
        done += ft8o_refine2_subtract(wf, tones, &f, &t, 1);
        done += ft8o_subtract(wf, tones, msgs[i].fHz + 3.0, msgs[i].tsec);
        f = msgs[i].fHz; t = (i == 0) ? 0.0001 : 2.45;            /* start sample 1; a signal that runs off the end of the buffer */
        done += ft8o_refine2_subtract(wf, tones, &f, &t, 1);
    }
    /* the local re-search: a mask shorter than / as long as / longer than the search range, then the configured search again */
    static uint8_t mask[5000];
    for (int i = 0; i < 5000; i++) mask[i] = (uint8_t)((i % 11) == 0);
    const int lens[3] = {100, cfg->f0_hi - cfg->f0_lo, 5000};
    for (int k = 0; k < 3; k++) {
        ft8o_set_search_mask(mask, lens[k]);
        ft8o_decode_frame(audio, cfg, cands, &nc, evlog, 4096, &nl, msgs, 256, &nm);
        done += nm;
    }
    ft8o_set_search_mask(NULL, 0);
    return done;
}

int main(int argc, char** argv) {
    static int16_t audio[FT8O_NSAMP];
    ft8o_config ref, ext;
    ft8o_default_config(&ref);
    ext = ref; ext.bp_iters_b = 30; ext.osd_triple = 12; ext.osd_max_hd = 34;
    memset(audio, 0, sizeof(audio));
    printf("silence: %d messages\n", run(audio, &ref));
    for (int i = 1; i < argc; i++) {
        int n = load(argv[i], audio);
        if (n < 0) { fprintf(stderr, "cannot read %s\n", argv[i]); return 2; }
        printf("%s: %d samples, %d messages (reference knobs), %d (extension knobs)\n", argv[i], n, run(audio, &ref), run(audio, &ext));
        printf("%s: %d subtractions\n", argv[i], run_sub(audio, &ref));
    }
    return 0;
}
