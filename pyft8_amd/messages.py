"""Host-side FT8 message layer: 77-bit payload -> text, call-hash table, frame packaging.

Behavioural contract = reference PyFT8/decoders.py:16-115 (unpack and friends), PyFT8/databases.py:8-26
(add_call_hashes) and PyFT8/receiver.py:51-66 (check_and_package).  The GPU decides *validity*
(csrc/ft8_dev.h: ft8_valid77); this module only renders strings and replays the hash-table side
effects in the reference's call order.  Written table-driven from the FT8 message layout.
"""
import time

from .ft8_tables import PFX1_MASK, PFX1_TRAP, PFX2_BITMAP

A37 = " 0123456789ABCDEFGHIJKLMNOPQRSTUVWXYZ"
A38 = A37 + "/"
A27 = " ABCDEFGHIJKLMNOPQRSTUVWXYZ"
NTOKENS, MAX22 = 2063592, 4194304
AP_NAMES = ("NoAP", "CQ", "RR73", "73", "RRR")
M64 = (1 << 64) - 1


class CallHashes:
    """(hash, nbits) -> callsign, nbits in {10, 12, 22}; last writer wins (databases.py:8-26)."""

    def __init__(self):
        self.by_hash = {}
        self.by_call = {}

    def clear(self):
        self.by_hash.clear()
        self.by_call.clear()

    def add(self, call):
        acc = 0
        for ch in (call + " " * 11)[:11]:
            acc = (acc * 38 + A38.find(ch)) & M64
        acc = (acc * 47055833459) & M64
        hs = [(acc >> (64 - nb), nb) for nb in (10, 12, 22)]
        for key in hs:
            self.by_hash[key] = call
        self.by_call[call] = hs

    def lookup(self, h, nb):
        return self.by_hash.get((h, nb), "...")


def _plausible(call):
    """decoders.py:107-115 without the file append."""
    if " " in call or len(call) < 3:
        return False
    a, b, c = call[0], call[1], call[2]
    if "A" <= a <= "Z" and (PFX1_MASK >> (ord(a) - 65)) & 1 and b.isdigit():
        if not ((PFX1_TRAP >> (ord(a) - 65)) & 1 and c.isdigit()):
            return True
    ia, ib = A37.find(a) - 1, A37.find(b) - 1
    return ia >= 0 and ib >= 0 and (PFX2_BITMAP[ia] >> ib) & 1 == 1 and c.isdigit()


def _call28_text(n28):
    v = n28 - NTOKENS - MAX22
    if v < 0:
        return "ZZ9ZZZ"          # python divmod / negative-index artefact of the reference at n28 = 6257895
    out = []
    for alphabet, size in ((A37, 37), (A37[1:], 36), ("0123456789", 10), (A27, 27), (A27, 27), (A27, 27))[::-1]:
        v, r = divmod(v, size)
        out.append(alphabet[r])
    return "".join(reversed(out)).strip()


def _field29(v29, i3, table):
    flag, n28 = v29 & 1, v29 >> 1
    if n28 < 3:
        return ("DE", "QRZ", "CQ")[n28]
    if n28 < 1004:
        return "CQ %03d" % (n28 - 3)
    if n28 < 21443:
        v, s = n28 - 1003, ""
        for _ in range(4):
            v, r = divmod(v, 27)
            s = A27[r] + s
        return "CQ " + s.strip()
    if n28 < NTOKENS + MAX22 - 1:
        return "<%s>" % table.lookup(n28 - NTOKENS, 22)
    call = _call28_text(n28)
    if not _plausible(call):
        return None
    if flag:
        call += "/P" if i3 == 2 else "/R"
        if call.endswith("/R") and call[0] not in "AKNW":
            return None
    table.add(call)
    return call


def _grid_or_report(g16):
    g15 = g16 & 0x7FFF
    if g15 < 32400:
        q, r = divmod(g15, 1800)
        s, r = divmod(r, 100)
        return chr(65 + q) + chr(65 + s) + "%02d" % r
    if g15 <= 32404:
        return ("", "", "RRR", "RR73", "73")[g15 - 32400]
    return ("R" if g16 >> 15 else "") + "%+03d" % (g15 - 32435)


def unpack(bits77, table):
    """77-bit int -> (call_a, call_b, extra) or None; mutates `table` exactly like the reference."""
    if not bits77:
        return None
    i3 = bits77 & 7
    body = bits77 >> 3
    if i3 in (1, 2):
        g16, cb, ca = body & 0xFFFF, (body >> 16) & 0x1FFFFFFF, (body >> 45) & 0x1FFFFFFF
        if g16 & 0x7FFF == 0:
            return None
        extra = _grid_or_report(g16)
        a = _field29(ca, i3, table)
        b = _field29(cb, i3, table)
        if a is None or b is None or extra == "":
            return None
        return (a, b, extra)
    if i3 == 4:
        cq, rrr, swap = body & 1, (body >> 1) & 3, (body >> 3) & 1
        n58, h12 = (body >> 4) & ((1 << 58) - 1), (body >> 62) & 0xFFF
        if bool(cq) == bool(rrr):
            return None
        first = "CQ" if cq else "<%s>" % table.lookup(h12, 12)
        s = ""
        for _ in range(12):
            n58, r = divmod(n58, 38)
            s = A38[r] + s
        s = s.strip()
        table.add(s)
        pair = (s, first) if swap else (first, s)
        return pair + (("", "RRR", "RR73", "73")[rrr],)
    return None


def decode_notes(rec):
    """'{source}_{AP}_{method}' + tweaks, formatted as the reference does (receiver.py:42,57,121,126,133,162)."""
    fine = rec["ipass"] >= 2
    meth = ("GOOD91 ", "LDPC5", "LDPC20", "OSD", "LDPC20_OSD")[rec["method"]]
    return ("fine" if fine else "grid") + "_" + AP_NAMES[rec["ap"]] + "_" + meth, tweaks_str(rec)


def tweaks_str(rec):
    if rec["ipass"] >= 2:
        return " t:%+03d f:%+03d" % (rec["ttweak"], rec["ftweak"])
    return "t:%+03d f:%+03d" % (0, 0)


_LAST_IPASS = {2: 0, 3: 1, 4: 1}     # status -> last ladder step taken (STOP_GRID_SD, STOP_COSTAS, STOP_FINE_SD)


def package_frame(rec, count, events, n_events, cyclestart_string="", band=None, odd_even=0, table=None, on_message=None):
    """Replay one frame's candidate records in the reference's order (receiver.py:389-398):
    per round all live candidates advance one ipass in llr_sd-descending (stable) order; CRC-passing
    unpack() calls update the hash table as they happen; first sighting of a message text is emitted.
    Returns the list of message dicts (reference receiver.py:61-64 keys)."""
    table = table if table is not None else CallHashes()
    rec = rec[:count]
    n_ev = min(int(n_events), len(events))
    per = {}
    for e in events[:n_ev]:
        per.setdefault((int(e["cand"]), int(e["ipass"])), []).append((int(e["slot"]), int(e["seq"]), (int(e["msg_hi"]) << 64) | int(e["msg_lo"])))
    last = []
    for r in rec:
        st = int(r["status"])
        last.append(int(r["ipass"]) if st == 1 else _LAST_IPASS.get(st, 7))
    out, seen = [], set()
    for rnd in range(8):
        live = [i for i in range(len(rec)) if last[i] >= rnd]
        if rnd == 1:
            live.sort(key=lambda i: float(rec[i]["grid_sd"]), reverse=True)
        elif rnd >= 2:
            live.sort(key=lambda i: float(rec[i]["fine_sd"]), reverse=True)
        for i in live:
            r = rec[i]
            decoded_here = int(r["status"]) == 1 and int(r["ipass"]) == rnd
            stop_key = None
            if decoded_here:
                m = int(r["method"])
                slot = int(r["ap"]) + (5 if m == 4 else 0)
                seq = 0 if m == 0 else (int(r["n_its"]) + 1 if m in (1, 2) else int(r["n_its"]))
                stop_key = (slot, seq)
            text = None
            done = set()
            for slot, seq, bits in sorted(per.get((i, rnd), [])):
                if stop_key is not None and (slot, seq) > stop_key:
                    break
                if (slot, seq) in done:
                    continue
                done.add((slot, seq))
                res = unpack(bits, table)
                if stop_key == (slot, seq):
                    text = res
            if decoded_here:
                if text is None:        # event log truncated: render at emit time
                    text = unpack((int(r["msg_hi"]) << 64) | int(r["msg_lo"]), table)
                if text is None:
                    continue
                msg_text = " ".join(text)
                if msg_text in seen:
                    continue
                seen.add(msg_text)
                fine = rnd >= 2
                tsec = int(r["h0_idx"]) / 25.0
                fHz = 3.125 * int(r["f0_idx"])
                if fine:
                    tsec = float(tsec + int(r["ttweak"]) / 200)
                    fHz = float(fHz + int(r["ftweak"]) / 16)
                snr = "%+03d" % int(r["snr_fine"] if fine else r["snr_grid"])
                notes, tw = decode_notes(r)
                m = {"band": band, "tsec": tsec, "fHz": fHz, "msg_tuple": text, "their_snr": snr,
                     "their_tx_cycle": odd_even,
                     "all_txt_format": f"{cyclestart_string} {snr} {(tsec - 0.6):4.1f} {fHz:4.0f} ~ {msg_text}",
                     "cyclestart_string": cyclestart_string, "decode_completed": time.time(), "tweaks": tw,
                     "decode_notes": notes + tw}
                out.append(m)
                if on_message is not None:
                    on_message(m)
    return out


def message_dicts(msgs, count, cyclestart_string="", band=None, odd_even=0, on_message=None):
    """Rows of the native packager (ft8rx_package_batch, _lib.MESSAGE_DTYPE) -> the reference's message dicts
    (receiver.py:57-65).  Same formatting as package_frame above."""
    out = []
    now = time.time()
    rows = msgs[:min(int(count), len(msgs))]
    if len(rows) == 0:
        return out
    # whole columns to Python lists first: field access on numpy structured scalars costs more than everything else here
    cols = [rows[k].tolist() for k in ("f", "h0_idx", "f0_idx", "ttweak", "ftweak", "snr", "ipass", "method", "ap", "fine")]
    for f3, h0, f0, tt, ft, sn, ipass, method, ap, fn in zip(*cols):
        text = tuple(x.decode() for x in f3)
        tsec = h0 / 25.0
        fHz = 3.125 * f0
        if fn:
            tsec = float(tsec + tt / 200)
            fHz = float(fHz + ft / 16)
        snr = "%+03d" % sn
        rec = {"ipass": ipass, "method": method, "ap": ap, "ttweak": tt, "ftweak": ft}
        notes, tw = decode_notes(rec)
        d = {"band": band, "tsec": tsec, "fHz": fHz, "msg_tuple": text, "their_snr": snr, "their_tx_cycle": odd_even,
             "all_txt_format": f"{cyclestart_string} {snr} {(tsec - 0.6):4.1f} {fHz:4.0f} ~ {' '.join(text)}",
             "cyclestart_string": cyclestart_string, "decode_completed": now, "tweaks": tw, "decode_notes": notes + tw}
        out.append(d)
        if on_message is not None:
            on_message(d)
    return out
