// host_messages.hpp -- native host message layer: 77-bit words -> strings, call hashes, duplicate filter (decoders.py:16-115, databases.py:10-26, receiver.py:51-66)
// Part of libft8rx.so; included by ft8rx.hip (single translation unit: the kernels share __constant__/__device__ tables).
#ifndef FT8RX_HOST_MESSAGES_HPP
#define FT8RX_HOST_MESSAGES_HPP

// ====================================================================================== host message layer (native)
// C++ twin of pyft8_amd/messages.py: 77-bit payload -> text (reference decoders.py:16-115), call-hash table
// (databases.py:8-26) and the per-frame replay of records/events in the reference's emit order with its duplicate
// filter (receiver.py:51-66, 389-398).  Pure host code, no HIP: frames are independent and are packaged by a pool
// of threads so that the Python surface is not the bottleneck behind ~26 k decoded frames/s.
#include <algorithm>
#include <thread>
#include <unordered_map>
namespace hostmsg {
static const char A37[] = " 0123456789ABCDEFGHIJKLMNOPQRSTUVWXYZ";
static const char A38[] = " 0123456789ABCDEFGHIJKLMNOPQRSTUVWXYZ/";
static const char A27[] = " ABCDEFGHIJKLMNOPQRSTUVWXYZ";
struct Hashes {
    std::unordered_map<uint64_t, std::string> m;                 // key = nbits << 32 | hash
    void add(const std::string& call) {
        uint64_t acc = 0;
        for (int i = 0; i < 11; i++) {
            char ch = i < (int)call.size() ? call[i] : ' ';
            const char* q = strchr(A38, ch);
            int64_t idx = (q && ch) ? (int64_t)(q - A38) : -1;
            acc = acc * 38 + (uint64_t)idx;
        }
        acc *= 47055833459ULL;
        const int nb[3] = {10, 12, 22};
        for (int k = 0; k < 3; k++) m[((uint64_t)nb[k] << 32) | (acc >> (64 - nb[k]))] = call;
    }
    std::string get(uint32_t h, int nb) const { auto it = m.find(((uint64_t)nb << 32) | h); return it == m.end() ? std::string("...") : it->second; }
};
static std::string strip(const std::string& t) {
    size_t a = 0, b = t.size();
    while (a < b && t[a] == ' ') a++;
    while (b > a && t[b - 1] == ' ') b--;
    return t.substr(a, b - a);
}
static bool plausible(const std::string& c) {
    if (c.size() < 3 || c.find(' ') != std::string::npos) return false;
    auto dig = [](char x) { return x >= '0' && x <= '9'; };
    auto a36 = [](char x) { return (x >= '0' && x <= '9') ? x - '0' : (x >= 'A' && x <= 'Z') ? x - 'A' + 10 : -1; };
    if (c[0] >= 'A' && c[0] <= 'Z' && ((FT8_PFX1_MASK >> (c[0] - 'A')) & 1u) && dig(c[1]))
        if (!(((FT8_PFX1_TRAP >> (c[0] - 'A')) & 1u) && dig(c[2]))) return true;
    int x0 = a36(c[0]), x1 = a36(c[1]);
    return x0 >= 0 && x1 >= 0 && ((FT8_PFX2[x0] >> x1) & 1ULL) && dig(c[2]);
}
static bool field29(uint32_t v29, int i3, Hashes& H, std::string& out) {
    const uint32_t flag = v29 & 1u, n28 = v29 >> 1;
    char buf[24];
    if (n28 < 3) { out = n28 == 0 ? "DE" : n28 == 1 ? "QRZ" : "CQ"; return true; }
    if (n28 < 1004) { snprintf(buf, sizeof buf, "CQ %03u", n28 - 3); out = buf; return true; }
    if (n28 < 21443) {
        uint32_t v = n28 - 1003; std::string t(4, ' ');
        for (int i = 3; i >= 0; i--) { t[i] = A27[v % 27]; v /= 27; }
        out = "CQ " + strip(t); return true;
    }
    if (n28 < 2063592u + 4194303u) { out = "<" + H.get(n28 - 2063592u, 22) + ">"; return true; }
    std::string call;
    int64_t v = (int64_t)n28 - (2063592 + 4194304);
    if (v < 0) call = "ZZ9ZZZ";                                   // negative-index artefact of the reference at n28 = 6257895
    else {
        char ch[7]; ch[6] = 0;
        ch[5] = A27[v % 27]; v /= 27; ch[4] = A27[v % 27]; v /= 27; ch[3] = A27[v % 27]; v /= 27;
        ch[2] = (char)('0' + v % 10); v /= 10; ch[1] = A37[1 + v % 36]; v /= 36; ch[0] = A37[v % 37];
        call = strip(ch);
    }
    if (!plausible(call)) return false;
    if (flag) {
        call += (i3 == 2) ? "/P" : "/R";
        if (i3 != 2 && !(call[0] == 'A' || call[0] == 'K' || call[0] == 'N' || call[0] == 'W')) return false;
    }
    H.add(call);
    out = call;
    return true;
}
// unpack(): true + 3 fields when the reference returns a tuple; mutates H exactly like the reference
static bool unpack(uint64_t lo, uint64_t hi, Hashes& H, std::string f[3]) {
    if (!lo && !hi) return false;
    const unsigned i3 = (unsigned)(lo & 7u);
    if (i3 == 1 || i3 == 2) {
        const uint32_t g16 = (uint32_t)((lo >> 3) & 0xFFFFu), cb = (uint32_t)((lo >> 19) & 0x1FFFFFFFu);
        const uint32_t ca = (uint32_t)(((lo >> 48) | (hi << 16)) & 0x1FFFFFFFu), g15 = g16 & 0x7FFFu;
        if (g15 == 0) return false;
        char g[16];
        if (g15 < 32400) { unsigned q = g15 / 1800, r = g15 % 1800; snprintf(g, sizeof g, "%c%c%02u", 'A' + q, 'A' + r / 100, r % 100); }
        else if (g15 <= 32404) { static const char* T5[5] = {"", "", "RRR", "RR73", "73"}; snprintf(g, sizeof g, "%s", T5[g15 - 32400]); }
        else snprintf(g, sizeof g, "%s%+03d", (g16 >> 15) ? "R" : "", (int)g15 - 32435);
        const bool oka = field29(ca, (int)i3, H, f[0]);
        const bool okb = field29(cb, (int)i3, H, f[1]);
        f[2] = g;
        return oka && okb && g[0] != 0;
    }
    if (i3 == 4) {
        const unsigned cq = (unsigned)((lo >> 3) & 1u), rrr = (unsigned)((lo >> 4) & 3u), swp = (unsigned)((lo >> 6) & 1u);
        uint64_t n58 = ((lo >> 7) | (hi << 57)) & ((1ULL << 58) - 1);
        const uint32_t h12 = (uint32_t)((hi >> 1) & 0xFFFu);
        if ((cq != 0) == (rrr != 0)) return false;
        std::string first = cq ? std::string("CQ") : "<" + H.get(h12, 12) + ">";
        std::string t(12, ' ');
        for (int i = 11; i >= 0; i--) { t[i] = A38[n58 % 38]; n58 /= 38; }
        t = strip(t);
        H.add(t);
        static const char* R4[4] = {"", "RRR", "RR73", "73"};
        f[0] = swp ? t : first; f[1] = swp ? first : t; f[2] = R4[rrr];
        return true;
    }
    return false;
}
struct Ev { int cand, ipass, slot, seq; uint64_t lo, hi; };
static int package_frame(const ft8rx_record* rec, int n, const ft8rx_event* ev, int nev, ft8rx_message* out, int cap) {
    std::vector<Ev> E; E.reserve((size_t)nev);
    for (int i = 0; i < nev; i++) E.push_back({ev[i].cand, ev[i].ipass, ev[i].slot, ev[i].seq, ev[i].msg_lo, ev[i].msg_hi});
    std::sort(E.begin(), E.end(), [](const Ev& a, const Ev& b) {
        if (a.cand != b.cand) return a.cand < b.cand; if (a.ipass != b.ipass) return a.ipass < b.ipass;
        if (a.slot != b.slot) return a.slot < b.slot; return a.seq < b.seq; });
    std::vector<int> last(n), order; order.reserve(n);
    for (int i = 0; i < n; i++) {
        const int st = rec[i].status;
        last[i] = st == FT8RX_ST_DECODED ? rec[i].ipass : st == FT8RX_ST_STOP_GRID_SD ? 0 : (st == FT8RX_ST_STOP_COSTAS || st == FT8RX_ST_STOP_FINE_SD) ? 1 : 7;
    }
    Hashes H; std::vector<std::string> seen; int nm = 0;
    for (int rnd = 0; rnd < 8; rnd++) {
        order.clear();
        for (int i = 0; i < n; i++) if (last[i] >= rnd) order.push_back(i);
        if (rnd == 1) std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return rec[a].grid_sd > rec[b].grid_sd; });
        else if (rnd >= 2) std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return rec[a].fine_sd > rec[b].fine_sd; });
        for (int i : order) {
            const ft8rx_record& r = rec[i];
            const bool here = r.status == FT8RX_ST_DECODED && r.ipass == rnd;
            int sslot = -1, sseq = -1;
            if (here) {
                const int m = r.method;
                sslot = r.ap + (m == FT8RX_M_LDPC_B_OSD ? 5 : 0);
                sseq = m == FT8RX_M_GOOD91 ? 0 : (m == FT8RX_M_LDPC_A || m == FT8RX_M_LDPC_B) ? r.n_its + 1 : r.n_its;
            }
            std::string f[3], got[3]; bool have = false;
            Ev key{i, rnd, -1, -1, 0, 0};
            auto it = std::lower_bound(E.begin(), E.end(), key, [](const Ev& a, const Ev& b) {
                if (a.cand != b.cand) return a.cand < b.cand; if (a.ipass != b.ipass) return a.ipass < b.ipass;
                if (a.slot != b.slot) return a.slot < b.slot; return a.seq < b.seq; });
            int pslot = -2, pseq = -2;
            for (; it != E.end() && it->cand == i && it->ipass == rnd; ++it) {
                if (here && (it->slot > sslot || (it->slot == sslot && it->seq > sseq))) break;
                if (it->slot == pslot && it->seq == pseq) continue;          // the same call logged twice
                pslot = it->slot; pseq = it->seq;
                const bool ok = unpack(it->lo, it->hi, H, f);
                if (here && it->slot == sslot && it->seq == sseq) { have = ok; if (ok) { got[0] = f[0]; got[1] = f[1]; got[2] = f[2]; } }
            }
            if (!here) continue;
            if (!have) { have = unpack(r.msg_lo, r.msg_hi, H, got); if (!have) continue; }    // event log truncated
            std::string text = got[0] + " " + got[1] + " " + got[2];
            if (std::find(seen.begin(), seen.end(), text) != seen.end()) continue;
            seen.push_back(text);
            if (nm < cap) {
                ft8rx_message& o = out[nm]; memset(&o, 0, sizeof(o));
                snprintf(o.f[0], 16, "%s", got[0].c_str()); snprintf(o.f[1], 16, "%s", got[1].c_str()); snprintf(o.f[2], 16, "%s", got[2].c_str());
                o.cand = (int16_t)i; o.f0_idx = r.f0_idx; o.h0_idx = r.h0_idx; o.ipass = r.ipass; o.ap = r.ap; o.method = r.method;
                const bool fine = rnd >= 2;
                o.fine = fine; o.snr = fine ? r.snr_fine : r.snr_grid; o.ttweak = fine ? r.ttweak : 0; o.ftweak = fine ? r.ftweak : 0;
            }
            nm++;
        }
    }
    return nm;
}
}  // namespace hostmsg

#endif
