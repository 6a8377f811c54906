// host_messages.hpp -- native host message layer: 77-bit words -> strings, call hashes, duplicate filter (decoders.py:16-115, databases.py:10-26, receiver.py:51-66)
// Part of libft8rx.so; included by ft8rx.hip.  Pure host C++ (needs only include/ft8rx.h and ft8_tables.h): it is also compiled on
// its own by g++ -fsanitize=address,undefined for the host sanitizer target (oracle/Makefile `asan`, tests/host_asan_driver.cpp).
#ifndef FT8RX_HOST_MESSAGES_HPP
#define FT8RX_HOST_MESSAGES_HPP

// ====================================================================================== host message layer (native)
// C++ twin of pyft8_amd/messages.py: 77-bit payload -> text (reference decoders.py:16-115), call-hash table
// (databases.py:8-26) and the per-frame replay of records/events in the reference's emit order with its duplicate
// filter (receiver.py:51-66, 389-398).  Pure host code, no HIP: frames are independent and are packaged by a pool
// of threads so that the Python surface is not the bottleneck behind ~26 k decoded frames/s.
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include <algorithm>
#include <atomic>
#include <mutex>
#include <string>
#include <vector>
#include <thread>
#include <unordered_map>
#include <condition_variable>
#include <functional>
#include <unistd.h>
namespace hostmsg {
// Persistent worker threads of the packaging entry points.  Creating and joining n std::threads per call was ~1 ms of the 1.2 ms a
// 256-frame batch took at 32 threads (and, with one process per GPU, tens of thousands of thread creations per second on the host).
// run(n, fn) executes fn on up to n threads -- the caller and pool threads 1 .. n - 1 (created on first need, detached, parked on a
// condition variable between calls); fn must hand out its own work (an atomic counter) and return when none is left.  The caller
// returns when its own fn has returned and every pool thread that ENTERED fn has left it: a pool thread that wakes up late (a loaded
// host) finds the run closed and is not waited for.  One run at a time per process; the state is never freed (threads may still be
// parked on it at exit) and is rebuilt in a forked child, whose copy of it has no threads behind it.
class HostPool {
    struct State {
        std::mutex run_m, m;
        std::condition_variable cv_go, cv_done;
        const std::function<void(int)>* job = nullptr;
        int n_threads = 0, job_n = 0, entered = 0, exited = 0;
        bool open = false;
        uint64_t gen = 0;
        pid_t pid = 0;
    };
    static State*& state() { static State* st = nullptr; return st; }
    static void worker(State* st, int idx) {
        uint64_t seen = 0;
        std::unique_lock<std::mutex> lk(st->m);
        for (;;) {
            st->cv_go.wait(lk, [&] { return st->gen != seen; });
            seen = st->gen;
            if (st->open && idx < st->job_n) {
                const std::function<void(int)>* j = st->job;
                st->entered++;
                lk.unlock();
                (*j)(idx);
                lk.lock();
                if (++st->exited == st->entered) st->cv_done.notify_one();
            }
        }
    }
public:
    static void run(int n, const std::function<void(int)>& fn) {
        if (n <= 1) { fn(0); return; }
        static std::mutex create_m;
        State* st;
        {
            std::lock_guard<std::mutex> g(create_m);
            if (!state() || state()->pid != getpid()) { state() = new State; state()->pid = getpid(); }
            st = state();
        }
        std::lock_guard<std::mutex> g(st->run_m);
        {
            std::unique_lock<std::mutex> lk(st->m);
            while (st->n_threads < n - 1) { const int idx = ++st->n_threads; std::thread(worker, st, idx).detach(); }
            st->job = &fn; st->job_n = n; st->entered = st->exited = 0; st->open = true; st->gen++;
        }
        st->cv_go.notify_all();
        fn(0);
        std::unique_lock<std::mutex> lk(st->m);
        st->open = false;                                           // nobody enters from here on
        st->cv_done.wait(lk, [&] { return st->exited == st->entered; });
        st->job = nullptr; st->job_n = 0;
    }
};
// optional reject log: the reference appends every call that fails simple_validate_call to 'rejected_callsigns.txt' in the
// working directory (decoders.py:114-115).  Off by default; ft8rx_set_reject_log(path) turns it on for the process.
static std::string g_reject_log;
static std::mutex g_reject_mu;
static std::atomic<bool> g_reject_on{false};      // the packaging threads' fast path; the path string itself is only read under the mutex
static void set_reject_log(const char* path) {
    std::lock_guard<std::mutex> lk(g_reject_mu);
    g_reject_log = path ? path : "";
    g_reject_on.store(!g_reject_log.empty(), std::memory_order_release);
}
static void log_reject(const std::string& call) {
    if (!g_reject_on.load(std::memory_order_acquire)) return;
    std::lock_guard<std::mutex> lk(g_reject_mu);
    if (g_reject_log.empty()) return;
    if (FILE* f = fopen(g_reject_log.c_str(), "a")) { fprintf(f, "%s\n", call.c_str()); fclose(f); }
}
static const char A37[] = " 0123456789ABCDEFGHIJKLMNOPQRSTUVWXYZ";
static const char A38[] = " 0123456789ABCDEFGHIJKLMNOPQRSTUVWXYZ/";
static const char A27[] = " ABCDEFGHIJKLMNOPQRSTUVWXYZ";
// The call-hash table (databases.py:8-26): key = (number of hash bits, hash) -> callsign, last writer wins.  An open-addressing
// table of fixed 24-byte entries: every valid callsign of every replayed unpack() call is entered under three keys, so a node-based
// map spent most of the message layer's time in its allocator.
struct Hashes {
    struct Entry { uint64_t key; char call[15]; uint8_t len; };   // len = 0xFF: empty slot (keys themselves may be any value)
    std::vector<Entry> tab;
    size_t used = 0;
    Hashes() { tab.resize(1024); clear(); }
    void clear() { for (auto& e : tab) e.len = 0xFF; used = 0; }
    size_t size() const { return used; }
    static size_t slot_of(uint64_t key, size_t cap) { return (size_t)((key * 0x9E3779B97F4A7C15ULL) >> 32) & (cap - 1); }
    void put(uint64_t key, const std::string& call) {
        if (2 * (used + 1) > tab.size()) {                        // keep the load factor <= 1/2
            std::vector<Entry> old; old.swap(tab);
            tab.resize(old.size() * 2); clear();
            for (const auto& e : old) if (e.len != 0xFF) put(e.key, std::string(e.call, e.len));
        }
        size_t i = slot_of(key, tab.size());
        while (tab[i].len != 0xFF && tab[i].key != key) i = (i + 1) & (tab.size() - 1);
        if (tab[i].len == 0xFF) used++;
        tab[i].key = key;
        tab[i].len = (uint8_t)(call.size() < sizeof(tab[i].call) ? call.size() : sizeof(tab[i].call));
        memcpy(tab[i].call, call.data(), tab[i].len);
    }
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
        for (int k = 0; k < 3; k++) put(((uint64_t)nb[k] << 32) | (acc >> (64 - nb[k])), call);
    }
    std::string get(uint32_t h, int nb) const {
        const uint64_t key = ((uint64_t)nb << 32) | h;
        size_t i = slot_of(key, tab.size());
        while (tab[i].len != 0xFF) { if (tab[i].key == key) return std::string(tab[i].call, tab[i].len); i = (i + 1) & (tab.size() - 1); }
        return std::string("...");
    }
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
static bool validate(const std::string& c) { const bool ok = plausible(c); if (!ok) log_reject(c); return ok; }
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
    if (!validate(call)) return false;
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
        if (g15 < 32400) { unsigned q = g15 / 1800, r = g15 % 1800; g[0] = (char)('A' + q); g[1] = (char)('A' + r / 100); g[2] = (char)('0' + (r % 100) / 10); g[3] = (char)('0' + r % 10); g[4] = 0; }
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
static bool ev_before(const Ev& a, const Ev& b) {          // the reference's call order: candidate, ladder step, attempt, order inside it
    if (a.cand != b.cand) return a.cand < b.cand;
    if (a.ipass != b.ipass) return a.ipass < b.ipass;
    if (a.slot != b.slot) return a.slot < b.slot;
    return a.seq < b.seq;
}
// -> number of messages written (<= cap).  *flags: FT8RX_PKG_MSG_TRUNCATED if more than cap messages were emitted.
// sparse = the records of a packed result buffer (include/ft8rx.h): only the candidates that decoded or logged an event, in candidate
// order, each carrying its candidate index in pad2.  The replay never looks at any other candidate, and a stable sort orders a
// subset exactly as it orders it inside the full list, so the messages are the ones the dense arrays give.
static int package_frame(const ft8rx_record* rec, int n, const ft8rx_event* ev, int nev, ft8rx_message* out, int cap, Hashes& H, int* flags,
                         bool sparse = false) {
    std::vector<Ev> E; E.reserve((size_t)nev);
    int16_t pos[FT8RX_MAX_CANDS];
    if (sparse) {
        for (int i = 0; i < FT8RX_MAX_CANDS; i++) pos[i] = -1;
        for (int i = 0; i < n; i++) if (rec[i].pad2 < (uint32_t)FT8RX_MAX_CANDS) pos[rec[i].pad2] = (int16_t)i;
    }
    for (int i = 0; i < nev; i++) {
        int c = ev[i].cand;
        if (sparse) { c = c < FT8RX_MAX_CANDS ? pos[c] : -1; if (c < 0) continue; }
        E.push_back({c, ev[i].ipass, ev[i].slot, ev[i].seq, ev[i].msg_lo, ev[i].msg_hi});
    }
    std::sort(E.begin(), E.end(), ev_before);
    std::vector<int> last(n), order; order.reserve(n);
    for (int i = 0; i < n; i++) {
        const int st = rec[i].status;
        last[i] = st == FT8RX_ST_DECODED ? rec[i].ipass : st == FT8RX_ST_STOP_GRID_SD ? 0 : (st == FT8RX_ST_STOP_COSTAS || st == FT8RX_ST_STOP_FINE_SD) ? 1 : 7;
    }
    std::vector<std::string> seen; int nm = 0;
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
            auto it = std::lower_bound(E.begin(), E.end(), key, ev_before);
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
                for (int k = 0; k < 3; k++) { const size_t L = got[k].size() < 15 ? got[k].size() : 15; memcpy(o.f[k], got[k].data(), L); }      // (o is zeroed: NUL-terminated)
                o.cand = (int16_t)(sparse ? (int)r.pad2 : i); o.f0_idx = r.f0_idx; o.h0_idx = r.h0_idx; o.ipass = r.ipass; o.ap = r.ap; o.method = r.method;
                const bool fine = rnd >= 2;
                o.fine = fine; o.snr = fine ? r.snr_fine : r.snr_grid; o.ttweak = fine ? r.ttweak : 0; o.ftweak = fine ? r.ftweak : 0;
            }
            nm++;
        }
    }
    if (nm > cap) { if (flags) *flags |= FT8RX_PKG_MSG_TRUNCATED; nm = cap; }
    return nm;
}

// CRC-14 of a 77-bit message, bit-serial (decoders.py:123-129)
static inline unsigned crc14_serial(uint64_t lo, uint64_t hi) {
    unsigned r = 0;
    for (int i = 0; i < 96; i++) {
        unsigned b = 0;
        if (i < 77) { int pos = 76 - i; b = (unsigned)((pos >= 64 ? (hi >> (pos - 64)) : (lo >> pos)) & 1u); }
        unsigned top = (r >> 13) & 1u;
        r = ((r << 1) & 0x3FFFu) | b;
        if (top) r ^= 0x2757u;
    }
    return r;
}
// 77-bit word -> 79 tones (transmitter.py:181-223 encode_bits77): CRC-14, LDPC(174,91) systematic encode, Gray map, Costas framing
static void encode_tones(uint64_t lo, uint64_t hi, uint8_t* t) {
    static const uint8_t costas[7] = {3, 1, 4, 0, 6, 5, 2}, gray[8] = {0, 1, 3, 2, 5, 6, 4, 7};
    hi &= 0x1FFFull;
    const unsigned crc = crc14_serial(lo, hi);
    uint64_t cw[3] = {0, 0, 0};                                    // codeword bit v at word v >> 6, bit v & 63
    for (int r = 0; r < 91; r++) {                                 // message bit r: 77 message bits (bit 76 first), then the CRC
        unsigned b;
        if (r < 77) { const int pos = 76 - r; b = (unsigned)((pos >= 64 ? (hi >> (pos - 64)) : (lo >> pos)) & 1u); }
        else b = (crc >> (13 - (r - 77))) & 1u;
        if (b) { cw[0] ^= FT8_G0[r][0]; cw[1] ^= FT8_G0[r][1]; cw[2] ^= FT8_G0[r][2]; }
    }
    for (int k = 0; k < 7; k++) { t[k] = costas[k]; t[36 + k] = costas[k]; t[72 + k] = costas[k]; }
    for (int sidx = 0; sidx < 58; sidx++) {
        unsigned v = 0;
        for (int b = 0; b < 3; b++) { const int bit = 3 * sidx + b; v = (v << 1) | (unsigned)((cw[bit >> 6] >> (bit & 63)) & 1ull); }
        t[(sidx < 29 ? 7 : 14) + sidx] = gray[v];
    }
}

// records/events of n_frames frames -> messages; table == nullptr: a fresh hash table per frame (frames spread over n_threads
// threads); table != nullptr: frames in order on the caller's thread, all sharing (and updating) that table.
static int package_batch(const ft8rx_record* records, const int32_t* counts, const ft8rx_event* events, const int32_t* event_counts,
                         int n_frames, int max_cands, ft8rx_message* out, int max_msgs, int32_t* out_counts, int n_threads,
                         Hashes* table, int32_t* flags) {
    if (!records || !counts || !events || !event_counts || !out || !out_counts || n_frames < 1 || max_cands < 1 || max_msgs < 1) return -1;
    if (n_threads < 1 || table) n_threads = 1;
    if (n_threads > n_frames) n_threads = n_frames;
    std::atomic<int> next{0};                                     // frames are handed out one by one: they differ in length
    auto work = [&](int) {
        Hashes local;                                             // one table per worker, emptied for every frame
        for (int f = next.fetch_add(1, std::memory_order_relaxed); f < n_frames; f = next.fetch_add(1, std::memory_order_relaxed)) {
            int fl = 0;
            int nev = event_counts[f];
            if (nev > FT8RX_EVENT_CAP) { nev = FT8RX_EVENT_CAP; fl |= FT8RX_PKG_EVENTS_TRUNCATED; }
            if (nev < 0) nev = 0;
            int n = counts[f] < 0 ? 0 : (counts[f] > max_cands ? max_cands : counts[f]);
            if (!table) local.clear();
            out_counts[f] = package_frame(records + (size_t)f * max_cands, n, events + (size_t)f * FT8RX_EVENT_CAP, nev,
                                          out + (size_t)f * max_msgs, max_msgs, table ? *table : local, &fl);
            if (flags) flags[f] = fl;
        }
    };
    HostPool::run(n_threads, work);
    return 0;
}
// The same from a packed result buffer (header | frame table | kept records | used events; include/ft8rx.h): frames
// [frame_lo, frame_lo + n_frames) -> out[n_frames][max_msgs].  Every offset is checked against `bytes` before it is followed.
static int package_packed(const void* packed, uint64_t bytes, int frame_lo, int n_frames, ft8rx_message* out, int max_msgs,
                          int32_t* out_counts, int n_threads, Hashes* table, int32_t* flags) {
    if (!packed || !out || !out_counts || n_frames < 1 || frame_lo < 0 || max_msgs < 1 || bytes < sizeof(ft8rx_packed_header)) return -1;
    ft8rx_packed_header hd; memcpy(&hd, packed, sizeof(hd));
    if (hd.magic != FT8RX_PACKED_MAGIC || hd.overflow || hd.bytes > bytes || hd.n_frames < 0 || hd.n_records < 0 || hd.n_events < 0 ||
        (int64_t)frame_lo + n_frames > hd.n_frames) return -1;
    if (hd.bytes != sizeof(hd) + (uint64_t)hd.n_frames * sizeof(ft8rx_packed_frame) + (uint64_t)hd.n_records * sizeof(ft8rx_record) +
                    (uint64_t)hd.n_events * sizeof(ft8rx_event)) return -1;
    const unsigned char* base = (const unsigned char*)packed;
    const ft8rx_packed_frame* tab = (const ft8rx_packed_frame*)(base + sizeof(hd));
    const ft8rx_record* recs = (const ft8rx_record*)(base + sizeof(hd) + (size_t)hd.n_frames * sizeof(ft8rx_packed_frame));
    const ft8rx_event* evs = (const ft8rx_event*)((const unsigned char*)recs + (size_t)hd.n_records * sizeof(ft8rx_record));
    for (int f = frame_lo; f < frame_lo + n_frames; f++) {
        const ft8rx_packed_frame& t = tab[f];
        const int ne = t.n_ev > FT8RX_EVENT_CAP ? FT8RX_EVENT_CAP : (t.n_ev < 0 ? 0 : t.n_ev);
        if (t.rec_off < 0 || t.ev_off < 0 || (int64_t)t.rec_off + t.n_rec > hd.n_records || (int64_t)t.ev_off + ne > hd.n_events ||
            t.n_rec > FT8RX_MAX_CANDS) return -1;
    }
    if (n_threads < 1 || table) n_threads = 1;
    if (n_threads > n_frames) n_threads = n_frames;
    std::atomic<int> next{0};
    auto work = [&](int) {
        Hashes local;
        for (int i = next.fetch_add(1, std::memory_order_relaxed); i < n_frames; i = next.fetch_add(1, std::memory_order_relaxed)) {
            const ft8rx_packed_frame& t = tab[frame_lo + i];
            int fl = 0, nev = t.n_ev < 0 ? 0 : t.n_ev;
            if (nev > FT8RX_EVENT_CAP) { nev = FT8RX_EVENT_CAP; fl |= FT8RX_PKG_EVENTS_TRUNCATED; }
            if (!table) local.clear();
            out_counts[i] = package_frame(recs + t.rec_off, t.n_rec, evs + t.ev_off, nev, out + (size_t)i * max_msgs, max_msgs,
                                          table ? *table : local, &fl, true);
            if (flags) flags[i] = fl;
        }
    };
    HostPool::run(n_threads, work);
    return 0;
}
// Multi-pass decoding (extension): append the messages of a later pass that are new for their frame.
// out[f][0 .. out_counts[f]) holds the messages so far; add[f][0 .. add_counts[f]) the later pass's; a message is new if its three
// text fields differ from every message the frame already has.  New ones are appended to out (pad[0] = pass_tag) and, untagged, to
// fresh[f] -- the list the next subtraction sweep works from.  drop_osd: ignore OSD decodes (ipass 5 / 6) of the later pass.
static void merge_messages(ft8rx_message* out, int32_t* out_counts, int max_out, const ft8rx_message* add, const int32_t* add_counts,
                           int max_add, int n_frames, int pass_tag, int drop_osd, ft8rx_message* fresh, int32_t* fresh_counts) {
    for (int f = 0; f < n_frames; f++) {
        ft8rx_message* o = out + (size_t)f * max_out;
        const ft8rx_message* a = add + (size_t)f * max_add;
        ft8rx_message* fr = fresh ? fresh + (size_t)f * max_add : nullptr;
        int n = out_counts[f] < 0 ? 0 : out_counts[f], nf = 0;
        const int na = add_counts[f] < max_add ? add_counts[f] : max_add;
        for (int i = 0; i < na; i++) {
            if (drop_osd && (a[i].method == FT8RX_M_OSD || a[i].method == FT8RX_M_LDPC_B_OSD)) continue;
            if (n >= max_out) break;
            bool seen = false;
            for (int j = 0; j < n && !seen; j++) seen = memcmp(o[j].f, a[i].f, sizeof(a[i].f)) == 0;
            if (seen) continue;
            o[n] = a[i]; o[n].pad[0] = (uint8_t)pass_tag; n++;
            if (fr) fr[nf] = a[i];
            nf++;
        }
        out_counts[f] = n;
        if (fresh_counts) fresh_counts[f] = nf;
    }
}
// Input of a subtraction sweep (multi-pass extension): every message of a frame with snr > min_snr, in emit order, as
// (tones of its codeword, the decoder's origin: fHz = 3.125 f0_idx (+ ftweak / 16), tsec = h0_idx / 25 (+ ttweak / 200) -- the values
// the message dict reports, receiver.py:166).  -> the largest per-frame count.
static int subtraction_list(const ft8rx_message* msgs, const int32_t* counts, int max_msgs, const ft8rx_record* records, int max_cands,
                            int n_frames, int min_snr, ft8rx_subsig* sigs, int max_sigs, int32_t* sig_counts) {
    int most = 0;
    for (int f = 0; f < n_frames; f++) {
        const ft8rx_message* m = msgs + (size_t)f * max_msgs;
        ft8rx_subsig* o = sigs + (size_t)f * max_sigs;
        int n = 0;
        const int nm = counts[f] < max_msgs ? counts[f] : max_msgs;
        for (int i = 0; i < nm && n < max_sigs; i++) {
            if ((int)m[i].snr <= min_snr || m[i].cand < 0 || m[i].cand >= max_cands) continue;
            const ft8rx_record& r = records[(size_t)f * max_cands + m[i].cand];
            memset(&o[n], 0, sizeof(o[n]));
            encode_tones(r.msg_lo, r.msg_hi, o[n].tones);
            o[n].fHz = 3.125 * (double)m[i].f0_idx + (m[i].fine ? (double)m[i].ftweak / 16.0 : 0.0);
            o[n].tsec = (double)m[i].h0_idx / 25.0 + (m[i].fine ? (double)m[i].ttweak / 200.0 : 0.0);
            n++;
        }
        sig_counts[f] = n;
        if (n > most) most = n;
    }
    return most;
}
}  // namespace hostmsg

#endif
