"""CPU-only tests of the product's host side: the C ABI library loads and exports every declared symbol,
the message layer / frame packaging replay reproduces the reference's dicts when fed oracle records, and the
N>1 gather path works over gloo with world_size 2.  No compute calls into the GPU library here."""
import ctypes
import json
import os
import re
import subprocess
import sys

import numpy as np
import pytest

from conftest import GOLDEN_FRAMES, ROOT, load_golden
from helpers import oracle_frame, records_from_oracle


def test_c_abi_exports_every_declared_symbol():
    from pyft8_amd import _lib
    hdr = open(os.path.join(ROOT, "include", "ft8rx.h")).read()
    names = sorted(set(re.findall(r"\b(ft8rx_[a-z0-9_]+)\s*\(", hdr)))
    assert len(names) >= 20
    L = _lib.lib()
    for n in names:
        assert hasattr(L, n), f"libft8rx.so does not export {n}"
    W = _lib.lib(wide=True)                                    # the wide-layout build of the same source (search ranges beyond 3 kHz)
    for n in names:
        assert hasattr(W, n), f"libft8rx_wide.so does not export {n}"
    info = [ctypes.c_int32() for _ in range(3)]
    W.ft8rx_build_info(*[ctypes.byref(v) for v in info])
    assert [v.value for v in info] == [1920, 96000, 1888]
    L.ft8rx_build_info(*[ctypes.byref(v) for v in info])
    assert [v.value for v in info] == [976, 49152, 960]
    # and nothing CPU-side pretends to be the product: create must fail without a GPU
    if L.ft8rx_device_count() == 0:
        with pytest.raises(_lib.Ft8rxError):
            _lib.Handle()


def test_merge_messages_native():
    """ft8rx_merge_messages (multi-pass extension, host only): new texts are appended with the pass tag and returned as the fresh
    list; texts the frame already has, OSD decodes when asked, and anything beyond the capacity are not."""
    from pyft8_amd import _lib
    def mk(rows, cap):
        a = np.zeros((len(rows), cap), _lib.MESSAGE_DTYPE)
        c = np.zeros(len(rows), np.int32)
        for f, msgs in enumerate(rows):
            for i, (txt, method) in enumerate(msgs):
                for k, w in enumerate(txt.split(" ")):
                    a[f, i]["f"][k] = w.encode()
                a[f, i]["method"] = method
            c[f] = len(msgs)
        return a, c
    out, oc = mk([[("CQ K1ABC FN42", 0), ("K1ABC W9XYZ EN37", 1)], [], [("CQ DL1AA JO62", 0)]], 4)
    add, ac = mk([[("K1ABC W9XYZ EN37", 2), ("W9XYZ K1ABC -05", 3), ("CQ PA5S JO21", 1), ("CQ PA5S JO21", 1)],
                  [("CQ G4ABC IO91", 4)], [("CQ DL1AA JO62", 1), ("A B C", 1), ("D E F", 1), ("G H I", 1), ("J K L", 1)]], 5)
    fresh, fc = _lib.merge_messages(out, oc, add, ac, 1, drop_osd=True)
    txt = lambda a, n: [b" ".join(r["f"]).decode() for r in a[:n]]
    assert txt(out[0], oc[0]) == ["CQ K1ABC FN42", "K1ABC W9XYZ EN37", "CQ PA5S JO21"]            # duplicate dropped, OSD (method 3) dropped
    assert txt(out[1], oc[1]) == [] and fc[1] == 0                                                   # method 4 = OSD on saved LLRs: dropped
    assert txt(out[2], oc[2]) == ["CQ DL1AA JO62", "A B C", "D E F", "G H I"] and fc[2] == 3         # capacity 4
    assert txt(fresh[0], fc[0]) == ["CQ PA5S JO21"] and int(out[0, 2]["pad"][0]) == 1 and int(fresh[0, 0]["pad"][0]) == 0
    fresh, fc = _lib.merge_messages(out, oc, add, ac, 2, drop_osd=False)
    assert txt(out[0], oc[0])[-1] == "W9XYZ K1ABC -05" and int(out[0, 3]["pad"][0]) == 2 and txt(out[1], oc[1]) == ["CQ G4ABC IO91"]


def test_subtraction_list_native_equals_numpy_twin():
    """ft8rx_subtraction_list (host only) against Receiver._subtraction_list_py on the message arrays of the golden frames."""
    from pyft8_amd import _lib
    from pyft8_amd.receiver import Receiver
    names = ["test_09", "synth_100000", "synth_200000"]
    recs = [records_from_oracle(oracle_frame(load_golden(n)[0])) for n in names]
    rec = np.stack([r[0] for r in recs]); cnt = np.array([r[1] for r in recs], np.int32)
    ev = np.zeros((len(names), _lib.EVENT_CAP), _lib.EVENT_DTYPE)
    for f, r in enumerate(recs):
        ev[f, :min(len(r[2]), _lib.EVENT_CAP)] = r[2][:_lib.EVENT_CAP]
    evc = np.array([r[3] for r in recs], np.int32)
    msgs, mcnt = _lib.package_batch(rec, cnt, ev, evc)
    for thr in (-10, -30, 3, -10.5):
        a, ac = _lib.subtraction_list(msgs, mcnt, rec, thr)
        b, bc = Receiver._subtraction_list_py(msgs, mcnt, rec, thr)
        assert np.array_equal(ac, bc) and ac.sum() > 0
        for f in range(len(names)):
            assert a[f, :ac[f]].tobytes() == b[f, :bc[f]].tobytes()


def test_record_layouts_match_header():
    from pyft8_amd import _lib
    assert _lib.RECORD_DTYPE.itemsize == 48 and _lib.EVENT_DTYPE.itemsize == 24
    assert ctypes.sizeof(_lib.Config) == 15 * 4          # include/ft8rx.h: 13 reference-derived fields + osd_triple, osd_max_hd
    plans = _lib.fft_plans()
    assert int(np.prod(plans["plan1920"])) == 1920 and int(np.prod(plans["plan3200"])) == 3200
    assert int(np.prod(plans["plan300"])) * int(np.prod(plans["plan320"])) == 96000


def test_config_from_receiver_kwargs():
    from pyft8_amd.receiver import config_from_kwargs
    c = config_from_kwargs()
    assert (c.f0_lo, c.f0_hi, c.h0_lo, c.h0_hi, c.max_cands, c.sync_score_min) == (32, 960, -37, 87, 200, 85.0)
    c = config_from_kwargs(sync_score_min=100, max_cands=150, bp_iters_b=30)        # pyft8.py:136 intent + extension knob
    assert (c.max_cands, c.sync_score_min, c.bp_iters_b) == (150, 100.0, 30)
    with pytest.raises(TypeError):                                                  # pyft8.py:137 passes `search_timerange`: the
        config_from_kwargs(search_timerange=[-2, 3])                                # reference's constructor rejects it, so does this one


@pytest.mark.parametrize("name", GOLDEN_FRAMES)
def test_package_frame_replays_reference_dicts(name):
    """Host replay (ordering, hash side effects, duplicate filter, dict formatting) fed with oracle records
    must give the reference's message dicts (minus decode_completed)."""
    from pyft8_amd import messages as M
    audio, gold, js = load_golden(name)
    rec, n, ev, nev = records_from_oracle(oracle_frame(audio))
    got = []
    msgs = M.package_frame(rec, n, ev, nev, cyclestart_string="700101_000015", band=None, odd_even=0, on_message=got.append)
    assert msgs == got and len(msgs) == len(js["messages"])
    for m, ref in zip(msgs, js["messages"]):
        for key, val in ref.items():
            assert (list(m[key]) if key == "msg_tuple" else m[key]) == val, (key, m[key], val)
        assert "decode_completed" in m


def test_package_frame_survives_truncated_event_log():
    from pyft8_amd import messages as M
    audio, gold, js = load_golden("test_08")
    rec, n, ev, nev = records_from_oracle(oracle_frame(audio))
    msgs = M.package_frame(rec, n, ev[:5], nev)
    assert sorted(" ".join(m["msg_tuple"]).replace("<...>", "#") for m in msgs) == \
        sorted(" ".join(m["msg_tuple"]).replace("<...>", "#") for m in js["messages"])


def test_shard_partition():
    from pyft8_amd.distributed import shard
    for n in (1, 7, 8, 256, 65536):
        for w in (1, 2, 4, 8):
            parts = [shard(n, r, w) for r in range(w)]
            assert parts[0][0] == 0 and sum(c for _, c in parts) == n
            assert all(parts[i][0] + parts[i][1] == parts[i + 1][0] for i in range(w - 1))
            assert max(c for _, c in parts) - min(c for _, c in parts) <= 1


_WORKER = r'''
import os, sys, json
import numpy as np
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, os.path.join(sys.argv[1], "tests")); sys.path.insert(0, os.path.join(sys.argv[1], "oracle"))
import torch.distributed as dist
from conftest import load_golden
from helpers import oracle_frame, records_from_oracle
from pyft8_amd import _lib, messages as M
from pyft8_amd.distributed import shard, gather_results
dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
names = ["synth_200000", "synth_100000", "synth_200000"]          # 3 frames over 2 ranks: ragged shards
start, count = shard(len(names), rank, world)
recs, cnts, evs, evcs = [], [], [], []
for nm in names[start:start + count]:
    rec, n, ev, nev = records_from_oracle(oracle_frame(load_golden(nm)[0]))
    e = np.zeros(_lib.EVENT_CAP, _lib.EVENT_DTYPE); e[:len(ev)] = ev[:_lib.EVENT_CAP]
    recs.append(rec); cnts.append(n); evs.append(e); evcs.append(nev)
out = gather_results(np.stack(recs), np.array(cnts, np.int32), np.stack(evs), np.array(evcs, np.int32), dst=0)
if rank == 0:
    rec, cnt, ev, evc = out
    res = [[" ".join(m["msg_tuple"]) for m in M.package_frame(rec[f], int(cnt[f]), ev[f], int(evc[f]))] for f in range(len(names))]
    json.dump(res, open(sys.argv[2], "w"))
else:
    assert out is None
dist.barrier(); dist.destroy_process_group()
'''


def test_gather_two_ranks_gloo(tmp_path):
    """world_size-2 run of the N>1 path on CPU (gloo): ragged shards, gather to rank 0, packaging."""
    script = tmp_path / "worker.py"
    script.write_text(_WORKER)
    out = tmp_path / "out.json"
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    subprocess.check_call([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                           "--master-addr", "127.0.0.1", "--master-port", "29617", str(script), ROOT, str(out)],
                          env=env, timeout=600, stdout=subprocess.DEVNULL, stderr=subprocess.STDOUT)
    res = json.load(open(out))
    want = [[" ".join(m["msg_tuple"]) for m in load_golden(nm)[2]["messages"]] for nm in ["synth_200000", "synth_100000", "synth_200000"]]
    assert res == want


def test_gather_forced_single_rank_gloo_roundtrips_bytes():
    """force=True runs the collectives in a one-rank group (the way the RCCL path is exercised on a one-GPU box): the gathered
    records / counts / used event rows are byte-identical to the inputs, and only the event columns in use travelled."""
    import torch.distributed as dist
    from pyft8_amd import _lib
    from pyft8_amd.distributed import gather_results
    rng = np.random.default_rng(5)
    B, mc = 6, 200
    rec = np.frombuffer(rng.bytes(B * mc * _lib.RECORD_DTYPE.itemsize), _lib.RECORD_DTYPE).reshape(B, mc).copy()
    ev = np.frombuffer(rng.bytes(B * _lib.EVENT_CAP * _lib.EVENT_DTYPE.itemsize), _lib.EVENT_DTYPE).reshape(B, _lib.EVENT_CAP).copy()
    cnt = rng.integers(0, mc, B).astype(np.int32)
    evc = np.array([0, 3, 41, 600, 7, 0], np.int32)              # one frame over the cap
    assert gather_results(rec, cnt, ev, evc)[2] is ev             # no group: pass-through
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:29631", rank=0, world_size=1)
    try:
        assert gather_results(rec, cnt, ev, evc)[2] is ev         # one rank, not forced: pass-through
        r2, c2, e2, ec2 = gather_results(rec, cnt, ev, evc, force=True)
        assert r2.tobytes() == rec.tobytes() and np.array_equal(c2, cnt) and np.array_equal(ec2, evc) and e2.shape == ev.shape
        for f in range(B):
            n = min(int(evc[f]), _lib.EVENT_CAP)
            assert e2[f, :n].tobytes() == ev[f, :n].tobytes()
        evc[3] = 50
        e3 = gather_results(rec, cnt, ev, evc, force=True)[2]
        assert e3[:, :50].tobytes() == ev[:, :50].tobytes() and not np.ascontiguousarray(e3[:, 50:]).view(np.uint8).any()       # columns past the used ones stayed home
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("name", GOLDEN_FRAMES)
def test_native_packager_matches_python_and_reference(name):
    """ft8rx_package_batch (C++, pure host) == messages.package_frame (Python) == the reference's dicts."""
    from pyft8_amd import _lib, messages as M
    audio, gold, js = load_golden(name)
    rec, n, ev, nev = records_from_oracle(oracle_frame(audio))
    evp = np.zeros(_lib.EVENT_CAP, _lib.EVENT_DTYPE)
    evp[:len(ev)] = ev[:_lib.EVENT_CAP]
    # two frames in one call (the second truncated to 5 events) to exercise batching / threads
    recs = np.stack([rec, rec]); evs = np.stack([evp, evp])
    msgs, cnt = _lib.package_batch(recs, np.array([n, n], np.int32), evs, np.array([nev, 5], np.int32), n_threads=2)
    got = M.message_dicts(msgs[0], cnt[0], cyclestart_string="700101_000015")
    py = M.package_frame(rec, n, ev, nev, cyclestart_string="700101_000015")
    assert len(got) == len(py) == len(js["messages"])
    for a, b, ref in zip(got, py, js["messages"]):
        for key, val in ref.items():
            assert (list(a[key]) if key == "msg_tuple" else a[key]) == val, (key, a[key], val)
            assert a[key] == b[key]
    trunc = sorted(" ".join(x.decode() for x in m["f"]).replace("<...>", "#") for m in msgs[1][:cnt[1]])
    assert trunc == sorted(" ".join(m["msg_tuple"]).replace("<...>", "#") for m in js["messages"])


def test_native_packager_flags_clamp_and_persistent_table(tmp_path):
    """counts never exceed the message capacity (truncation is flagged, not silent), an overflowed event log is flagged
    and warned about, and a persistent call-hash table carries hashed calls from one frame to the next (databases.py:8)."""
    import warnings
    from pyft8_amd import _lib, messages as M
    audio, gold, js = load_golden("test_09")
    rec, n, ev, nev = records_from_oracle(oracle_frame(audio))
    evp = np.zeros(_lib.EVENT_CAP, _lib.EVENT_DTYPE); evp[:len(ev)] = ev[:_lib.EVENT_CAP]
    recs, evs = rec[None], evp[None]
    nmsg = len(js["messages"])
    # capacity smaller than the message list: clamped + flagged + warned
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        msgs, cnt, flags = _lib.package_batch(recs, np.array([n], np.int32), evs, np.array([nev], np.int32), max_msgs=5, return_flags=True)
    assert cnt[0] == 5 and msgs.shape == (1, 5) and flags[0] & _lib.PKG_MSG_TRUNCATED
    assert any(issubclass(x.category, _lib.Ft8rxTruncationWarning) for x in w)
    # default capacity = record capacity: cannot truncate
    msgs, cnt, flags = _lib.package_batch(recs, np.array([n], np.int32), evs, np.array([nev], np.int32), return_flags=True)
    assert cnt[0] == nmsg and flags[0] == 0 and msgs.shape[1] == recs.shape[1]
    # event count beyond the log capacity: flagged
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        _, cnt2, flags = _lib.package_batch(recs, np.array([n], np.int32), evs, np.array([_lib.EVENT_CAP + 7], np.int32), return_flags=True)
    assert flags[0] & _lib.PKG_EVENTS_TRUNCATED and any(issubclass(x.category, _lib.Ft8rxTruncationWarning) for x in w)
    # persistent table: test_09 prints "<...> OR18OSB RR73" (OR18OSB defines its hash there); a second pass over the same frame with
    # the SAME table must resolve nothing new for standard calls but keeps the table filled; a fresh table per frame forgets
    tab = _lib.CallHashTable()
    assert len(tab) == 0
    m1, c1 = _lib.package_batch(recs, np.array([n], np.int32), evs, np.array([nev], np.int32), table=tab)
    assert len(tab) > 30                                   # 3 keys (10/12/22 bits) per call heard
    texts1 = [" ".join(x.decode() for x in m["f"]) for m in m1[0][:c1[0]]]
    assert texts1 == [" ".join(m["msg_tuple"]) for m in js["messages"]]
    # a frame that refers to OR18OSB by hash only: i3 = 1 word with a hashed first call
    tab2 = M.CallHashes(); tab2.add("OR18OSB")
    h22 = [k for k in tab2.by_hash if k[1] == 22][0][0]
    import oracle as O
    n28 = 2063592 + h22
    # "<OR18OSB> G4ABC IO91": build from the pieces of a golden standard message (call_b / grid of the first golden pack entry)
    d = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "messages.json")))
    w0 = int(d["pack"][0]["bits77"], 16)
    word = (w0 & ((1 << 48) - 1)) | ((n28 << 1) << 48)
    word = (word & ~7) | 1
    r2 = np.zeros_like(rec); r2[0] = rec[0]; r2[0]["status"] = 1; r2[0]["ipass"] = 0; r2[0]["method"] = 0; r2[0]["ap"] = 0
    r2[0]["msg_lo"] = word & (2 ** 64 - 1); r2[0]["msg_hi"] = word >> 64
    e2 = np.zeros((1, _lib.EVENT_CAP), _lib.EVENT_DTYPE)
    fresh, cf = _lib.package_batch(r2[None], np.array([1], np.int32), e2, np.array([0], np.int32))
    kept, ck = _lib.package_batch(r2[None], np.array([1], np.int32), e2, np.array([0], np.int32), table=tab)
    assert cf[0] == ck[0] == 1
    assert fresh[0][0]["f"][0] == b"<...>" and kept[0][0]["f"][0] == b"<OR18OSB>"
    tab.clear(); assert len(tab) == 0
    tab.add("OR18OSB"); assert len(tab) == 3
    # optional reject log (decoders.py:114-115): off by default, one line per rejected call when a path is set
    log = tmp_path / "rejected_callsigns.txt"
    bad = (w0 & ~(((1 << 29) - 1) << 48)) | ((((2063592 + 4194304 + 5) << 1)) << 48)      # call_a = an implausible 28-bit call
    r3 = r2.copy(); r3[0]["msg_lo"] = bad & (2 ** 64 - 1); r3[0]["msg_hi"] = bad >> 64
    _lib.package_batch(r3[None], np.array([1], np.int32), e2, np.array([0], np.int32))
    assert not log.exists()
    _lib.set_reject_log(str(log))
    try:
        _, c3 = _lib.package_batch(r3[None], np.array([1], np.int32), e2, np.array([0], np.int32))
    finally:
        _lib.set_reject_log(None)
    assert c3[0] == 0 and log.exists() and len(log.read_text().split()) >= 1


def test_native_packager_throughput():
    import time
    from pyft8_amd import _lib
    audio, gold, js = load_golden("synth_000000")
    rec, n, ev, nev = records_from_oracle(oracle_frame(audio))
    evp = np.zeros(_lib.EVENT_CAP, _lib.EVENT_DTYPE); evp[:len(ev)] = ev
    B = 256
    recs = np.stack([rec] * B); evs = np.stack([evp] * B)
    cnt = np.full(B, n, np.int32); evc = np.full(B, nev, np.int32)
    _lib.package_batch(recs, cnt, evs, evc, n_threads=1)
    t0 = time.perf_counter()
    msgs, mc = _lib.package_batch(recs, cnt, evs, evc, n_threads=1)
    dt = time.perf_counter() - t0
    assert (mc == len(js["messages"])).all()
    assert B / dt > 3000, f"native packager only {B / dt:.0f} frames/s on one thread"


def test_native_packager_worker_pool_thread_counts_and_concurrent_callers():
    """The packaging entry points run on a persistent worker pool (host_messages.hpp: HostPool): the result does not depend on the thread
    count, the pool grows and shrinks between calls, callers on several Python threads (ctypes releases the GIL) are serialised, and a
    child process that imported after a fork-free spawn builds its own pool (covered by the gloo tests, which package in every rank)."""
    import threading
    from pyft8_amd import _lib
    audio, gold, js = load_golden("synth_000000")
    rec, n, ev, nev = records_from_oracle(oracle_frame(audio))
    evp = np.zeros(_lib.EVENT_CAP, _lib.EVENT_DTYPE); evp[:len(ev)] = ev
    B = 97
    recs = np.stack([rec] * B); evs = np.stack([evp] * B)
    cnt = np.full(B, n, np.int32); evc = np.full(B, nev, np.int32)
    cnt[::7] = 0; evc[::5] = 0                                     # uneven frames: the pool hands frames out one by one
    want, wc = _lib.package_batch(recs, cnt, evs, evc, n_threads=1)
    for nt in (2, 32, 3, 64, 1, 8):
        got, gc_ = _lib.package_batch(recs, cnt, evs, evc, n_threads=nt)
        assert np.array_equal(gc_, wc) and got.tobytes() == want.tobytes(), nt
    # written in place into the arrays of an earlier call: the same counts and the same used messages (slots beyond the counts are stale)
    held = _lib.package_batch(recs, cnt[::-1].copy(), evs, evc, n_threads=8, return_flags=True)
    again = _lib.package_batch(recs, cnt, evs, evc, n_threads=8, return_flags=True, out=held)
    assert again[0] is held[0] and np.array_equal(again[1], wc)
    assert all(again[0][f, :wc[f]].tobytes() == want[f, :wc[f]].tobytes() for f in range(B))
    with pytest.raises(_lib.Ft8rxError):
        _lib.package_batch(recs[:5], cnt[:5], evs[:5], evc[:5], return_flags=True, out=held)
    errs = []

    def caller(k):
        try:
            for i in range(40):
                got, gc_ = _lib.package_batch(recs, cnt, evs, evc, n_threads=(4, 16, 7, 32)[(k + i) % 4])
                assert np.array_equal(gc_, wc) and got.tobytes() == want.tobytes()
        except Exception as e:                                      # noqa: BLE001
            errs.append(repr(e))
    ths = [threading.Thread(target=caller, args=(k,)) for k in range(4)]
    for t in ths:
        t.start()
    for t in ths:
        t.join(timeout=120)
    assert not errs and not any(t.is_alive() for t in ths), errs


def test_native_packager_worker_pool_survives_fork(tmp_path):
    """A forked child has the parent's pool object but none of its threads: the pool notices the new pid and starts its own
    (os.fork in a process of its own: the pytest process has other threads)."""
    script = tmp_path / "fork_pool.py"
    script.write_text(f'''
import os, sys, signal
import numpy as np
sys.path.insert(0, {ROOT!r})
from pyft8_amd import _lib
B = 64
rec = np.zeros((B, 8), _lib.RECORD_DTYPE); ev = np.zeros((B, _lib.EVENT_CAP), _lib.EVENT_DTYPE)
cnt = np.zeros(B, np.int32); evc = np.zeros(B, np.int32)
want = _lib.package_batch(rec, cnt, ev, evc, n_threads=8)[1].tobytes()          # the parent's pool exists now
pid = os.fork()
if pid == 0:
    signal.alarm(30)                                                            # a child waiting for threads it does not have dies here
    ok = _lib.package_batch(rec, cnt, ev, evc, n_threads=8)[1].tobytes() == want
    os._exit(0 if ok else 3)
_, status = os.waitpid(pid, 0)
assert _lib.package_batch(rec, cnt, ev, evc, n_threads=8)[1].tobytes() == want  # the parent's pool is untouched
sys.exit(0 if os.WIFEXITED(status) and os.WEXITSTATUS(status) == 0 else 4)
''')
    r = subprocess.run([sys.executable, str(script)], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, (r.returncode, r.stderr[-2000:])


def test_ragged_and_invalid_frames():
    """Host-side input handling: short frames are padded with silence, bad input raises Ft8rxError (no asserts, no UB)."""
    from pyft8_amd import _lib
    from pyft8_amd.receiver import frames_from_ragged, _as_frames
    a = frames_from_ragged([np.ones(10, np.int16), np.arange(5), np.zeros(0, np.int16)])
    assert a.shape == (3, _lib.NSAMP) and a.dtype == np.int16
    assert a[0, :10].tolist() == [1] * 10 and not a[0, 10:].any() and a[1, :5].tolist() == [0, 1, 2, 3, 4] and not a[2].any()
    full = np.arange(2 * _lib.NSAMP, dtype=np.int64).reshape(2, -1).astype(np.int16)
    assert _as_frames(full) is not None and np.array_equal(_as_frames(full), full)
    assert _as_frames(full[0]).shape == (1, _lib.NSAMP)
    assert _as_frames(np.zeros((0, _lib.NSAMP), np.int16)).shape == (0, _lib.NSAMP)
    for bad in ([np.zeros(_lib.NSAMP + 1, np.int16)], [np.zeros((2, 3), np.int16)], [np.array([1.5])], [np.array([40000])]):
        with pytest.raises(_lib.Ft8rxError):
            frames_from_ragged(bad)


def test_messages_py_unpack_matches_reference_golden():
    """pyft8_amd.messages.unpack + CallHashes against the reference's own unpack() run in sequence (tests/golden/messages.json)."""
    d = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "messages.json")))
    from pyft8_amd import messages as M, decoders
    tab = M.CallHashes()
    for u in d["unpack"]:
        m = M.unpack(int(u["bits77"], 16), tab)
        assert (None if m is None else " ".join(m)) == u["result"], u
    decoders.call_hashes.clear()                  # module-level surface of the reference (decoders.unpack + databases.call_hashes)
    for u in d["unpack"][:500]:
        m = decoders.unpack(int(u["bits77"], 16))
        assert (None if m is None else " ".join(m)) == u["result"], u


def test_native_tone_encoder_matches_reference_transmitter():
    """ft8rx_encode_tones (host, no GPU): CRC-14 + LDPC encode + Gray + Costas == the reference transmitter's encode_bits77 on the
    220 messages of tests/golden/messages.json."""
    from pyft8_amd import _lib
    d = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "messages.json")))
    words = [int(p["bits77"], 16) for p in d["pack"]]
    t = _lib.encode_tones([w & (2 ** 64 - 1) for w in words], [w >> 64 for w in words])
    assert ["".join(map(str, row)) for row in t] == [p["tones"] for p in d["pack"]]


def test_oracle_subtract_matches_reference_golden():
    """SURVEY 8f-4 primitive on the CPU: the oracle's restatement of Receiver.subtract_signal vs the reference run in isolation
    (tests/golden/subtract.npz, oracle/gen_golden_subtract.py): < 1e-5 of the audio RMS at 16384 sampled positions per case."""
    import oracle as O
    from pyft8_amd import synth
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "subtract.npz"))
    for ci in range(len(g["recipes"])):
        idx, ns, lo, hi = g["recipes"][ci]
        audio = synth.make_frame(int(idx), n_signals=int(ns), snr_range=(lo, hi)).astype(np.float32)
        done = [O.subtract(audio, t, f, ts) for t, f, ts in zip(g[f"c{ci}_tones"], g[f"c{ci}_fHz"], g[f"c{ci}_tsec"])]
        assert all(done) == (ci != 3)
        assert np.abs(audio[g["pos"]] - g[f"c{ci}_after"]).max() < 1e-5 * g[f"c{ci}_stats"][0]
        assert abs(audio.astype(np.float64).std() - g[f"c{ci}_stats"][1]) < 1e-3


def test_frames_from_wav_reads_the_reference_fixture_recordings(tmp_path):
    """receiver.frames_from_wav: the reference's two fixture recordings (mono int16 12 kHz, 15 s) come back as the golden's samples;
    longer recordings are cut into cycles, the last one zero-padded; other formats are rejected."""
    import wave
    from conftest import load_golden
    from pyft8_amd import _lib
    from pyft8_amd.receiver import frames_from_wav
    gdir = os.path.join(os.path.dirname(__file__), "golden")
    for name in ("test_08", "test_09"):
        fr = frames_from_wav(os.path.join(gdir, name + ".wav"))
        assert fr.shape == (1, 180000) and fr.dtype == np.int16 and np.array_equal(fr[0], load_golden(name)[0])
    long = np.arange(400000, dtype=np.int32).astype(np.int16)
    p = str(tmp_path / "long.wav")
    with wave.open(p, "wb") as w:
        w.setnchannels(1); w.setsampwidth(2); w.setframerate(12000); w.writeframes(long.tobytes())
    fr = frames_from_wav(p)
    assert fr.shape == (3, 180000) and np.array_equal(fr[1], long[180000:360000]) and np.array_equal(fr[2, :40000], long[360000:]) and not fr[2, 40000:].any()
    assert frames_from_wav(p, cycle_offset_s=30.0).shape == (1, 180000)
    with wave.open(p, "wb") as w:
        w.setnchannels(2); w.setsampwidth(2); w.setframerate(12000); w.writeframes(long[:1000].tobytes())
    with pytest.raises(_lib.Ft8rxError, match="mono"):
        frames_from_wav(p)


def test_bench_rank_placement_helpers():
    """bench.py's per-rank CPU placement: whole cores per rank, disjoint slices, readable ranges; never fatal."""
    import bench
    assert bench._fmt_cpus([0, 1, 2, 3, 8, 9, 11]) == "0-3,8-9,11"
    assert bench._cpulist("0-3,8,10-11\n") == [0, 1, 2, 3, 8, 10, 11]
    cpus = sorted(os.sched_getaffinity(0))
    parts = [bench._split_whole_cores(cpus, k, 2) for k in range(2)]
    assert all(parts) and (len(cpus) < 2 or not set(parts[0]) & set(parts[1]))
    keep = os.sched_getaffinity(0)
    try:
        info = bench.place_rank(0, 2, lambda r: None)                       # no PCI address known: slices of the allowed CPUs
        assert info["n_cpus"] >= 1 and "slice 1/2" in info["how"] and os.sched_getaffinity(0) == set(parts[0])
        info = bench.place_rank(1, 2, lambda r: "0000:ff:1f.7")             # an address sysfs does not know: same fallback
        assert "slice 2/2" in info["how"]
        bad = bench.place_rank(0, 1, lambda r: 1 / 0)                       # placement is an optimisation: errors are reported, not raised
        assert bad["how"].startswith("unpinned (ZeroDivisionError")
    finally:
        os.sched_setaffinity(0, keep)


# --------------------------------------------------------------------------------------- packed results (multi-GPU gather format)
def _dense_goldens(names):
    from pyft8_amd import _lib
    recs, cnts, evs, evcs = [], [], [], []
    for nm in names:
        rec, n, ev, nev = records_from_oracle(oracle_frame(load_golden(nm)[0]))
        e = np.zeros(_lib.EVENT_CAP, _lib.EVENT_DTYPE)
        e[:len(ev)] = ev[:_lib.EVENT_CAP]
        recs.append(rec); cnts.append(n); evs.append(e); evcs.append(nev)
    return np.stack(recs), np.array(cnts, np.int32), np.stack(evs), np.array(evcs, np.int32)


def test_osd_dpp_stages_keep_their_wait_states():
    """k_osd's sort network reads partner lanes through DPP operands inside inline-assembly blocks (csrc/kernels/osd.hpp); a VALU write
    of a DPP source needs two wait states before the read, and the compiler's hazard recognizer does not look inside inline asm
    (ADVICE r5).  tools/dpp_hazard_check.py scans the disassembly of both shipped builds: every one of the 360 DPP instructions per
    kernel keeps its distance -- a rebuild with another hipcc that schedules a producer right in front of an asm block fails here."""
    import importlib.util
    from conftest import ROOT
    from pyft8_amd import _lib
    spec = importlib.util.spec_from_file_location("dpp_hazard_check", os.path.join(ROOT, "tools", "dpp_hazard_check.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    if not os.path.exists(os.path.join(m.LLVM, "llvm-objdump")):
        pytest.skip("llvm-objdump not installed")
    for lib in (_lib.LIB_PATH, _lib.LIB_PATH_WIDE):
        n, bad = m.scan(lib)
        assert n >= 360 and not bad, (lib, n, bad[:3])


def test_host_layer_with_more_than_256_candidates():
    """max_cands is open-ended (receiver.py:311-313, 366-367): record arrays wider than 256 candidates are packaged by the copy of the
    host layer inside libft8rx_wide.so (FT8RX_MAX_CANDS = 2048).  Oracle records of one frame at max_cands = 600 / sync_score_min = 30:
    the native packager, the packed form and the Python twin all render the oracle's messages."""
    import oracle as O
    from pyft8_amd import _lib, synth, messages as M
    from pyft8_amd.receiver import config_from_kwargs
    cfg = config_from_kwargs(sync_score_min=30, max_cands=600)
    assert cfg.max_cands == 600 and config_from_kwargs(max_cands=10**12).max_cands == 928      # never more candidates than f0 bins
    audio = synth.make_frame(64100, n_signals=60, snr_range=(-14.0, 6.0))
    r = O.decode_frame(audio, O.default_config(**_lib.fft_plans(), sync_score_min=30.0, max_cands=600))
    rec, n, ev, nev = records_from_oracle(r, max_cands=600)
    assert 256 < n <= 600 and nev <= _lib.EVENT_CAP
    e = np.zeros((1, _lib.EVENT_CAP), _lib.EVENT_DTYPE)
    e[0, :nev] = ev[:nev]
    dense_in = (rec[None], np.array([n], np.int32), e, np.array([nev], np.int32))
    out, oc = _lib.package_batch(*dense_in)
    want = [" ".join(m["msg_tuple"]) for m in r["msgs"]]
    got = [" ".join(x.decode() for x in out[0, i]["f"] if x) for i in range(int(oc[0]))]
    assert got == want and len(want) > 10
    assert any(int(out[0, i]["cand"]) >= 256 for i in range(int(oc[0]))) or n > 256           # candidates beyond slot 255 take part
    buf = _lib.pack_results(*dense_in)
    pk = _lib.Packed(buf)
    assert int(pk.header["max_cands"]) == 600
    out2, oc2 = _lib.package_packed(buf)
    assert out2.tobytes() == out.tobytes() and np.array_equal(oc, oc2)
    twin = M.package_frame(rec, n, e[0], nev)
    assert [" ".join(m["msg_tuple"]) for m in twin] == want


def test_packed_results_render_the_same_messages_as_the_dense_arrays():
    """include/ft8rx.h "packed results": header | frame table | kept records | used events.  The numpy twin of the pack kernels
    (_lib.pack_results) on oracle records of the golden frames: ft8rx_package_packed == ft8rx_package_batch byte for byte (whole
    buffer and a sub-range), expand() puts every kept record back at its candidate position, every decoded candidate is kept, and
    malformed buffers are refused."""
    from pyft8_amd import _lib
    rec, cnt, ev, evc = _dense_goldens(["synth_200000", "test_09", "synth_100000", "test_08"])
    evc[3] = 700                                                   # an overflowed log keeps its raw count; only 512 entries travel
    buf = _lib.pack_results(rec, cnt, ev, evc)
    pk = _lib.Packed(buf)
    assert pk.n_frames == 4 and pk.nbytes == len(buf) and int(pk.header["max_cands"]) == 200
    assert pk.nbytes < 0.5 * (rec.nbytes + ev.nbytes)              # the point of the format
    dense = _lib.package_batch(rec, cnt, ev, evc, return_flags=True)
    packed = _lib.package_packed(buf, return_flags=True)
    assert dense[0].tobytes() == packed[0].tobytes() and np.array_equal(dense[1], packed[1]) and np.array_equal(dense[2], packed[2])
    assert packed[2][3] & _lib.PKG_EVENTS_TRUNCATED and int(dense[1].sum()) > 40
    sub = _lib.package_packed(buf, 1, 2)
    assert sub[0].tobytes() == dense[0][1:3].tobytes()
    r2, c2, e2, ec2 = pk.expand()
    assert np.array_equal(c2, cnt) and np.array_equal(ec2, evc)
    for f in range(4):
        kept = pk.frame(f)[0]["pad2"]
        assert np.all(np.diff(kept.astype(int)) > 0)                                 # candidate order
        assert r2[f, kept].tobytes() == rec[f, kept].tobytes()
        dec = np.nonzero(rec[f, :cnt[f]]["status"] == _lib.ST_DECODED)[0]
        assert set(dec) <= set(kept.tolist())
        n = min(int(evc[f]), _lib.EVENT_CAP)
        assert e2[f, :n].tobytes() == ev[f, :n].tobytes() and set(ev[f, :n]["cand"].tolist()) <= set(kept.tolist())
    # a NaN llr_sd anywhere in a frame keeps all of its candidates (stable-sort order of a subset is only defined for ordered keys)
    rec_nan = rec.copy()
    rec_nan[0, 3]["grid_sd"] = np.nan
    pkn = _lib.Packed(_lib.pack_results(rec_nan, cnt, ev, evc))
    assert int(pkn.frames[0]["n_rec"]) == int(cnt[0]) and int(pkn.frames[1]["n_rec"]) == int(pk.frames[1]["n_rec"])
    # malformed input: bad magic, cut short, overflow flag, offsets beyond the runs
    L = _lib.lib()
    out = np.zeros((4, 200), _lib.MESSAGE_DTYPE); oc = np.zeros(4, np.int32)
    L.ft8rx_package_packed.argtypes = [ctypes.c_void_p, ctypes.c_uint64, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_int,
                                       ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]

    def call(b, lo=0, n=4):
        return L.ft8rx_package_packed(b.ctypes.data, ctypes.c_uint64(len(b)), lo, n, out.ctypes.data, 200, oc.ctypes.data, 1, None, None)
    assert call(buf) == 0
    bad = buf.copy(); bad[0] ^= 1
    assert call(bad) == -1
    assert call(buf[:len(buf) - 8].copy()) == -1 and call(buf, 2, 3) == -1
    bad = buf.copy(); bad.view(np.uint8)[28:32].view(np.int32)[0] = 1                 # overflow flag
    assert call(bad) == -1
    bad = buf.copy(); bad[32:48].view(_lib.PACKED_FRAME_DTYPE)[0]["rec_off"] = 10 ** 6
    assert call(bad) == -1
    with pytest.raises(_lib.Ft8rxError):
        _lib.Packed(buf[:16])


_PACKED_WORKER = r'''
import ctypes, json, os, sys, types
import numpy as np
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, os.path.join(sys.argv[1], "tests")); sys.path.insert(0, os.path.join(sys.argv[1], "oracle"))
import torch.distributed as dist
from pyft8_amd import _lib
from pyft8_amd.distributed import shard, PackedGather


class FakeHandle:
    """Stands in for _lib.Handle on a box without a GPU: `batches` are dense result arrays; "fetching" batch k packs it (numpy twin of
    the pack kernels) into the buffer of slot k % 2 that PackedGather registered, exactly where the kernels would have written."""
    def __init__(self, batches):
        self.cfg = types.SimpleNamespace(max_cands=200)
        self.batches, self.k, self.bufs = batches, -1, None
    def pinned_bytes(self, n):
        return np.zeros(n, np.uint8)
    def set_packed_output(self, p0, p1, cap):
        self.bufs = None if not p0 else [np.frombuffer((ctypes.c_uint8 * cap).from_address(p), np.uint8) for p in (p0, p1)]
    def fetch(self):
        self.k += 1
        b = _lib.pack_results(*self.batches[self.k])
        self.bufs[self.k % 2][:len(b)] = b
    def packed_results(self):
        hdr = self.bufs[self.k % 2][:32].view(_lib.PACKED_HEADER_DTYPE)[0]
        return self.k % 2, {n: int(hdr[n]) for n in _lib.PACKED_HEADER_DTYPE.names}


dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
pool = np.load(sys.argv[2])
total, nb = int(sys.argv[4]), 3                                 # `total` frames per batch over the ranks (uneven), nb batches
start, count = shard(total, rank, world)
grow = len(sys.argv) > 5 and sys.argv[5] == "grow"
straggler = len(sys.argv) > 5 and sys.argv[5] == "straggler"
if straggler:
    nb = 12
def batch(k):                                                   # global frame g of batch k = pool frame (g + 5 k) % n
    idx = [(start + i + 5 * k) % len(pool["cnt"]) for i in range(count)]
    if grow and k == 0:
        idx = [0] * count                                       # the smallest frame only: the later batches need > 2x its bytes
    return pool["rec"][idx], pool["cnt"][idx], pool["ev"][idx], pool["evc"][idx]
h = FakeHandle([batch(k) for k in range(nb)])
if grow:
    PackedGather.MIN_ROW = 1024                                 # rank 0's receive rows start at twice the first batch's largest part
g = PackedGather(h, shard(total, 0, world)[1], dst=0, repeat=2 if grow else 1, depth=16 if straggler else 4)
got = []
import time
t_loop = time.perf_counter()
for k in range(nb):                                             # the bench loop's order: fetch k, submit k, (collect k - 1)
    if straggler and rank == 1:
        time.sleep(0.05)                                        # one slow rank: nobody but rank 0's collect() may wait for it
    h.fetch()
    g.submit()
    if straggler:
        continue
    if k > 0:
        parts = g.collect()
        if rank == 0:
            parts = parts[:world]                               # (repeat = 2 delivers every part twice)
            got.append([[" ".join(x.decode() for x in m["f"]) for m in ms[:n]] for p in parts for ms, n in zip(*_lib.package_packed(p))])
            assert [p.n_frames for p in parts] == [shard(total, r, world)[1] for r in range(world)]
        else:
            assert parts is None
t_loop = time.perf_counter() - t_loop
if straggler:
    # every rank's own submit loop, before anybody drains: the senders other than the straggler (and rank 0, whose ring is deep enough)
    # did not wait for it; then the batches arrive complete and in order
    times = [None] * world
    dist.all_gather_object(times, t_loop)
    assert times[1] >= 0.05 * nb and all(times[r] < 0.5 * times[1] for r in range(world) if r != 1), times
    # submit() never waits for the straggler's 50 ms: its typical cost is far below that (the median; one scheduling hiccup of a loaded
    # CPU box may exceed it once), and all of a rank's submits together stay below the straggler's delay of two batches
    assert sorted(g.seconds)[len(g.seconds) // 2] < 0.02 and sum(g.seconds) < 0.1 * nb, g.seconds
    for k in range(nb):
        parts = g.collect()
        if rank == 0:
            got.append([[" ".join(x.decode() for x in m["f"]) for m in ms[:n]] for p in parts for ms, n in zip(*_lib.package_packed(p))])
    if rank == 0:
        json.dump(got, open(sys.argv[3], "w"))
    g.close()
    dist.barrier(); dist.destroy_process_group()
    sys.exit(0)
parts = g.drain()
if rank == 0:
    assert len(parts) == world * (2 if grow else 1)
    if grow:
        rows = g.row_history                                    # the receive sets were re-allocated while parts were in flight
        assert len(rows) >= 2 and rows[-1] > rows[0] > 0, rows
        assert all(a.buf.tobytes() == b.buf.tobytes() for a, b in zip(parts[:world], parts[world:]))
    parts = parts[:world]
    got.append([[" ".join(x.decode() for x in m["f"]) for m in ms[:n]] for p in parts for ms, n in zip(*_lib.package_packed(p))])
    json.dump(got, open(sys.argv[3], "w"))
g.close()
dist.barrier(); dist.destroy_process_group()
'''


@pytest.mark.parametrize("world,total,mode", [(2, 5, ""), (8, 8 * 2 + 3, ""), (3, 7, "grow"), (4, 9, "straggler")])
def test_packed_gather_gloo_uneven_shards(tmp_path, world, total, mode):
    """PackedGather over gloo at world sizes 2 and 8 (the rank-count-dependent paths: uneven shard() blocks, byte counts that differ
    per rank, point-to-point sends announced through the store, several batches in flight) on CPU with a stand-in handle that packs oracle records: rank 0 ends up with every
    rank's frames in shard order, batch after batch, and renders the golden messages from the packed form.  "grow": the first batch is
    small, so rank 0's receive buffers (sized from the byte counts seen) are re-allocated while a gather is in flight; repeat = 2.
    "straggler": rank 1 sleeps 50 ms before every batch -- no rank but the straggler itself notices (there is no rendezvous: each rank
    waits for its own sends only, rank 0's ring holds 16 batches), and all twelve batches still arrive complete and in order."""
    from pyft8_amd import _lib
    names = ["synth_200000", "test_09", "synth_100000", "test_08", "synth_000000"]
    rec, cnt, ev, evc = _dense_goldens(names)
    np.savez(tmp_path / "pool.npz", rec=rec, cnt=cnt, ev=ev, evc=evc)
    script = tmp_path / "worker.py"
    script.write_text(_PACKED_WORKER)
    out = tmp_path / "out.json"
    subprocess.check_call([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr", "127.0.0.1",
                           "--master-port", str(29640 + world), str(script), ROOT, str(tmp_path / "pool.npz"), str(out), str(total), mode],
                          env=dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="1"), timeout=900, stdout=subprocess.DEVNULL, stderr=subprocess.STDOUT)
    got = json.load(open(out))
    want = [[" ".join(m["msg_tuple"]) for m in load_golden(nm)[2]["messages"]] for nm in names]
    assert len(got) == (12 if mode == "straggler" else 3)
    for k, batch in enumerate(got):
        assert batch == [want[0 if (mode == "grow" and k == 0) else (g + 5 * k) % len(names)] for g in range(total)], k


def test_every_kernel_exists_once():
    """libft8rx.so is linked from two translation units (ft8rx.hip; ft8rx_ilp.hip = the FFT kernels under the ILP scheduler): each
    kernel must exist in exactly one of the two device code objects, the FFT kernels in the second."""
    import shutil
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import kernel_resources as KR
    from pyft8_amd import _lib
    if not os.path.exists(os.path.join(KR.LLVM, "llvm-objdump")) or not shutil.which("c++filt"):
        pytest.skip("no LLVM binutils on this box")
    for path in (_lib.LIB_PATH, _lib.LIB_PATH_WIDE):
        ks = KR.kernels(path)
        names = [k[1] for k in ks]
        assert len(names) == len(set(names)) and len(names) > 30, sorted(n for n in names if names.count(n) > 1)
        unit = {k[1]: k[0] for k in ks}
        assert {n for n, u in unit.items() if u == 1} == {"k_fine", "k_spectrogram", "k_hop_spectrum"}
        assert all(unit[n] == 0 for n in ("k_bp", "k_osd", "k_sync", "k_cyc_a", "k_pack_write", "k_refine3"))
