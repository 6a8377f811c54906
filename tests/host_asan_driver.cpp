// host_asan_driver.cpp -- host sanitizer target for the product's pure-host C++.  TEST INFRASTRUCTURE.
// Compiles pyft8_amd/csrc/host_messages.hpp on its own (g++ -fsanitize=address,undefined; no HIP) and drives the native message
// layer -- unpack / call hashes / ordered replay / duplicate filter / tone encoder -- with (a) a records+events dump of a real frame
// written by the test (argv[1]: n, nev, then the raw ft8rx_record[n] and ft8rx_event[nev] bytes) and (b) random 77-bit words,
// multi-threaded, with fresh and with persistent hash tables, including truncating capacities.
//   make -C oracle asan && oracle/_build/asan_host [dump.bin]
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <random>
#include "../include/ft8rx.h"
#include "../pyft8_amd/csrc/ft8_tables.h"
#include "../pyft8_amd/csrc/host_messages.hpp"

int main(int argc, char** argv) {
    std::mt19937_64 rng(12345);
    const int MC = 200;
    std::vector<ft8rx_record> rec; std::vector<ft8rx_event> ev; int n = 0, nev = 0;
    if (argc > 1) {
        FILE* f = fopen(argv[1], "rb");
        if (!f) { fprintf(stderr, "cannot read %s\n", argv[1]); return 2; }
        int32_t hdr[2];
        if (fread(hdr, 4, 2, f) != 2) return 2;
        n = hdr[0]; nev = hdr[1];
        rec.resize(MC); ev.resize(FT8RX_EVENT_CAP);
        if ((int)fread(rec.data(), sizeof(ft8rx_record), n, f) != n || (int)fread(ev.data(), sizeof(ft8rx_event), nev, f) != nev) return 2;
        fclose(f);
    }
    // (a) the dumped frame, 8 copies over 4 threads, then sequentially against one persistent table, then with a tiny capacity
    long total = 0;
    if (n) {
        const int B = 8;
        std::vector<ft8rx_record> R((size_t)B * MC); std::vector<ft8rx_event> E((size_t)B * FT8RX_EVENT_CAP);
        std::vector<int32_t> cnt(B, n), evc(B, nev), oc(B), fl(B);
        for (int b = 0; b < B; b++) { std::copy(rec.begin(), rec.end(), R.begin() + (size_t)b * MC); std::copy(ev.begin(), ev.end(), E.begin() + (size_t)b * FT8RX_EVENT_CAP); }
        std::vector<ft8rx_message> out((size_t)B * MC);
        if (hostmsg::package_batch(R.data(), cnt.data(), E.data(), evc.data(), B, MC, out.data(), MC, oc.data(), 4, nullptr, fl.data())) return 3;
        for (int b = 0; b < B; b++) { if (oc[b] != oc[0] || fl[b]) return 4; total += oc[b]; }
        hostmsg::Hashes H;
        if (hostmsg::package_batch(R.data(), cnt.data(), E.data(), evc.data(), B, MC, out.data(), MC, oc.data(), 4, &H, fl.data())) return 3;
        evc[3] = FT8RX_EVENT_CAP + 100; cnt[5] = MC + 50;                      // overflowing counts are clamped and flagged
        if (hostmsg::package_batch(R.data(), cnt.data(), E.data(), evc.data(), B, MC, out.data(), 3, oc.data(), 2, nullptr, fl.data())) return 3;
        if (!(fl[3] & FT8RX_PKG_EVENTS_TRUNCATED) || !(fl[0] & FT8RX_PKG_MSG_TRUNCATED) || oc[0] != 3) return 5;
        {   // multi-pass merge: nothing of the same list is new; a list with one changed text is; capacity is respected
            std::vector<ft8rx_message> base((size_t)B * MC), add((size_t)B * MC), fresh((size_t)B * MC);
            std::vector<int32_t> bc(B), ac(B), fc(B);
            if (hostmsg::package_batch(R.data(), std::vector<int32_t>(B, n).data(), E.data(), std::vector<int32_t>(B, nev).data(), B, MC, base.data(), MC, bc.data(), 2, nullptr, fl.data())) return 3;
            add = base; ac = bc;
            hostmsg::merge_messages(base.data(), bc.data(), MC, add.data(), ac.data(), MC, B, 1, 0, fresh.data(), fc.data());
            for (int b = 0; b < B; b++) if (fc[b] != 0 || bc[b] != ac[b]) return 6;
            if (ac[0] > 0) {
                add[0].f[0][0] ^= 1;
                hostmsg::merge_messages(base.data(), bc.data(), MC, add.data(), ac.data(), MC, B, 2, 1, fresh.data(), fc.data());
                const bool osd = add[0].method == FT8RX_M_OSD || add[0].method == FT8RX_M_LDPC_B_OSD;
                if (fc[0] != (osd ? 0 : 1)) return 7;
                bc[1] = MC;                                                   // full list: nothing may be appended
                add[(size_t)MC].f[0][0] ^= 1;
                hostmsg::merge_messages(base.data(), bc.data(), MC, add.data(), ac.data(), MC, B, 3, 0, nullptr, nullptr);
                if (bc[1] != MC) return 8;
            }
        }
        {   // subtraction list of the multi-pass extension: bounded by max_sigs, tolerant of out-of-range candidate indices
            std::vector<ft8rx_message> ms((size_t)B * MC); std::vector<int32_t> mc(B);
            if (hostmsg::package_batch(R.data(), std::vector<int32_t>(B, n).data(), E.data(), std::vector<int32_t>(B, nev).data(), B, MC, ms.data(), MC, mc.data(), 2, nullptr, fl.data())) return 3;
            std::vector<ft8rx_subsig> sg((size_t)B * 4); std::vector<int32_t> sc(B);
            if (mc[0] > 0) ms[0].cand = 30000;
            const int most = hostmsg::subtraction_list(ms.data(), mc.data(), MC, R.data(), MC, B, -30, sg.data(), 4, sc.data());
            if (most > 4 || sc[0] > 4) return 9;
        }
        {   // packed results (multi-GPU gather format): a packed copy of the same frames renders the same messages; cut, corrupted and
            // random-offset buffers are refused or survive without touching memory outside the buffer
            std::vector<unsigned char> pk;
            {
                std::vector<ft8rx_packed_frame> tab(B);
                std::vector<ft8rx_record> pr; std::vector<ft8rx_event> pe;
                for (int b = 0; b < B; b++) {
                    std::vector<char> keep(MC, 0);
                    for (int i = 0; i < n; i++) keep[i] = rec[i].status == FT8RX_ST_DECODED;
                    for (int i = 0; i < nev; i++) if (ev[i].cand < n) keep[ev[i].cand] = 1;
                    tab[b].rec_off = (int32_t)pr.size(); tab[b].ev_off = (int32_t)pe.size(); tab[b].n_cand = (uint16_t)n; tab[b].n_ev = nev;
                    int k = 0;
                    for (int i = 0; i < n; i++) if (keep[i]) { ft8rx_record r = rec[i]; r.pad2 = (uint32_t)i; pr.push_back(r); k++; }
                    tab[b].n_rec = (uint16_t)k;
                    pe.insert(pe.end(), ev.begin(), ev.begin() + nev);
                }
                ft8rx_packed_header hd; memset(&hd, 0, sizeof hd);
                hd.magic = FT8RX_PACKED_MAGIC; hd.n_frames = B; hd.n_records = (int32_t)pr.size(); hd.n_events = (int32_t)pe.size(); hd.max_cands = MC;
                hd.bytes = sizeof(hd) + sizeof(ft8rx_packed_frame) * B + sizeof(ft8rx_record) * pr.size() + sizeof(ft8rx_event) * pe.size();
                pk.resize(hd.bytes);
                unsigned char* q = pk.data();
                memcpy(q, &hd, sizeof hd); q += sizeof hd;
                memcpy(q, tab.data(), sizeof(ft8rx_packed_frame) * B); q += sizeof(ft8rx_packed_frame) * B;
                memcpy(q, pr.data(), sizeof(ft8rx_record) * pr.size()); q += sizeof(ft8rx_record) * pr.size();
                memcpy(q, pe.data(), sizeof(ft8rx_event) * pe.size());
            }
            std::vector<ft8rx_message> dense((size_t)B * MC), packed((size_t)B * MC);
            std::vector<int32_t> dc(B), pc(B);
            if (hostmsg::package_batch(R.data(), std::vector<int32_t>(B, n).data(), E.data(), std::vector<int32_t>(B, nev).data(), B, MC, dense.data(), MC, dc.data(), 3, nullptr, fl.data())) return 3;
            if (hostmsg::package_packed(pk.data(), pk.size(), 0, B, packed.data(), MC, pc.data(), 3, nullptr, fl.data())) return 10;
            if (dc != pc || memcmp(dense.data(), packed.data(), sizeof(ft8rx_message) * dense.size())) return 11;
            if (hostmsg::package_packed(pk.data(), pk.size() - 1, 0, B, packed.data(), MC, pc.data(), 1, nullptr, nullptr) != -1) return 12;     // cut short
            if (hostmsg::package_packed(pk.data(), pk.size(), 1, B, packed.data(), MC, pc.data(), 1, nullptr, nullptr) != -1) return 12;         // frames beyond the batch
            for (int t = 0; t < 2000; t++) {                                    // corrupted headers / frame tables: -1 or a clean run, never a stray access
                std::vector<unsigned char> bad(pk);
                const size_t where = rng() % (sizeof(ft8rx_packed_header) + sizeof(ft8rx_packed_frame) * B);
                bad[where] ^= (unsigned char)(1u << (rng() % 8));
                if (t % 3 == 0) bad[rng() % (sizeof(ft8rx_packed_header) + sizeof(ft8rx_packed_frame) * B)] = (unsigned char)rng();
                hostmsg::package_packed(bad.data(), bad.size(), 0, B, packed.data(), MC, pc.data(), 2, nullptr, fl.data());
            }
            for (int t = 0; t < 300; t++) {                                     // corrupted records / events (candidate indices, counts in pad2): contents are untrusted too
                std::vector<unsigned char> bad(pk);
                const size_t lo = sizeof(ft8rx_packed_header) + sizeof(ft8rx_packed_frame) * B;
                for (int r = 0; r < 8; r++) bad[lo + rng() % (bad.size() - lo)] = (unsigned char)rng();
                hostmsg::package_packed(bad.data(), bad.size(), 0, B, packed.data(), MC, pc.data(), 2, nullptr, fl.data());
            }
        }
        printf("frame dump: %d candidates, %d events, %d messages per frame\n", n, nev, (int)(total / B));
    }
    // (b) random words through unpack (all i3 values, hashed calls, boundaries) and the encoder
    hostmsg::Hashes H; std::string f[3]; long ok = 0; uint8_t tones[79]; unsigned acc = 0;
    for (int i = 0; i < 200000; i++) {
        uint64_t lo = rng(), hi = rng() & 0x1FFF;
        if (i % 3 == 0) lo = (lo & ~7ull) | 1;
        if (i % 5 == 0) lo = (lo & ~7ull) | 4;
        if (i % 7 == 0) { lo &= 0xFFFFFFFFFFFFull; hi = 0; }
        ok += hostmsg::unpack(lo, hi, H, f);
        if (i % 16 == 0) { hostmsg::encode_tones(lo, hi, tones); acc += tones[40]; }
    }
    printf("random words: %ld unpacked, %zu hash keys, tone checksum %u\n", ok, H.size(), acc);
    return 0;
}
