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

static int run(const int16_t* audio, const ft8o_config* cfg) {
    static ft8o_cand cands[256]; static ft8o_event log[4096]; static ft8o_msg msgs[256];
    int32_t nc = 0, nl = 0, nm = 0;
    ft8o_decode_frame(audio, cfg, cands, &nc, log, 4096, &nl, msgs, 256, &nm);
    return nm;
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
    }
    return 0;
}
