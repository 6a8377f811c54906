"""-m gpu tests of what the 8-GPU box will run cold: the N > 1 bench path (one process per rank, launched exactly as the driver
launches it) and the BASELINE config-3 / config-4 workloads at their full per-GPU size, on the one GPU a test box has.

  * two ranks on device 0 over gloo: the whole multi-rank flow of bench.py (sharding, barriers, max-over-ranks timing, record
    gather, one JSON line from rank 0);
  * the same with --backend nccl: RCCL refuses two ranks on one device ("Duplicate GPU"), in which case the test is skipped --
    on a multi-GPU box it runs the device-resident RCCL gather for real;
  * `python bench.py --gpus 2` WITHOUT a launcher must start the launcher itself -- never a mislabelled 1-GPU run;
  * config 3: one rank's 8192-frame shard (~41 GB of workspaces) decodes, chunked-vs-whole and small-batch digests agree;
  * config 4: 2048 frames of <= 10 signals at -24..-20 dB with OSD order 3 + distance gate: runs, truth-based yield and false
    decodes bounded, a sample of frames identical to the oracle with the same knobs."""
import hashlib
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

import oracle as O
from conftest import ROOT

pytestmark = pytest.mark.gpu


def _port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _launch(nproc, extra, timeout=1500):
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={nproc}", "--master-addr", "127.0.0.1",
           "--master-port", str(_port()), os.path.join(ROOT, "bench.py"), "--gpus", str(nproc)] + extra
    return subprocess.run(cmd, capture_output=True, text=True, timeout=timeout, cwd=ROOT, env=dict(os.environ, MASTER_ADDR="127.0.0.1"))


def _one_line(out):
    lines = [l for l in out.stdout.splitlines() if l.strip().startswith("{")]
    assert len(lines) == 1, (out.stdout[-2000:], out.stderr[-2000:])
    return json.loads(lines[0])


SMALL = ["--frames", "32", "--steps", "2", "--warmup", "1", "--no-host-entry", "--min-seconds", "0"]


def test_two_ranks_one_gpu_gloo():
    out = _launch(2, ["--backend", "gloo"] + SMALL)
    assert out.returncode == 0, out.stderr[-3000:]
    d = _one_line(out)
    assert d["n_gpus"] == 2 and d["steps"] == 2 and d["scaling"] == "weak" and d["cpu_baseline"] is None
    assert d["config"]["frames_per_gpu"] == 32
    assert "packed results of 2 ranks gathered to rank 0 over gloo inside every timed step" in d["config"]["gather"], d["config"]["gather"]
    assert d["per_rank"]["gather"]["ok"] and 1000 < d["per_rank"]["gather"]["bytes_per_frame"] < 8000, d["per_rank"]["gather"]
    # whole-job value = frames of BOTH ranks / max-over-ranks time
    assert abs(d["value"] - 2 * 32 * 2 / (d["ms_per_step"] * 2 * 1e-3)) < 1e-6 * d["value"]
    assert d["value"] > 500


def test_rccl_gather_path_two_ranks_or_forced_single_rank():
    """The RCCL branch of bench.py (nccl process group; every timed step ends with PackedGather.submit: all_gather of the byte counts,
    gather of the packed device buffers, one D2H per rank on rank 0, on a side stream): two ranks where the box has two GPUs; on a
    one-GPU box (RCCL refuses two ranks on one device) the same collectives run in a ONE-rank nccl group (--force-gather) -- no skip
    either way."""
    import torch
    if torch.cuda.device_count() >= 2:
        out = _launch(2, ["--backend", "nccl"] + SMALL)
        assert out.returncode == 0, (out.stderr + out.stdout)[-3000:]
        d = _one_line(out)
        assert d["n_gpus"] == 2 and "packed results of 2 ranks gathered to rank 0 over nccl" in d["config"]["gather"], d["config"]["gather"]
        assert d["per_rank"]["gather"]["ok"]
        assert len(d["per_rank"]["frames_per_s"]) == 2 and len(d["per_rank"]["placement"]) == 2
    else:
        out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--backend", "nccl", "--force-gather",
                              "--no-cpu-baseline"] + SMALL, capture_output=True, text=True, timeout=900, cwd=ROOT,
                             env={k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")})
        assert out.returncode == 0, (out.stderr + out.stdout)[-3000:]
        d = _one_line(out)
        assert "packed results of 1 rank (forced) gathered to rank 0 over nccl inside every timed step" in d["config"]["gather"], d["config"]["gather"]
        g = d["per_rank"]["gather"]
        assert g["ok"] and g["submit_ms_per_step"] >= 0 and 1000 < g["bytes_per_frame"] < 8000 and d["per_rank"]["placement"][0]["how"], g


_NCCL_ONE_RANK = r'''
import os, sys
import numpy as np
sys.path.insert(0, sys.argv[1])
import torch, torch.distributed as dist
torch.cuda.set_device(0)
dist.init_process_group("nccl", init_method="tcp://127.0.0.1:" + sys.argv[2], rank=0, world_size=1, device_id=torch.device("cuda", 0))
from pyft8_amd import _lib
from pyft8_amd.distributed import gather_results_device, gather_results
B = 320                                             # > 1 MB of event log: the two-step (used columns only) result copy
h = _lib.Handle(max_frames=B)
ptr = h.staging_ptr()
h.synth_frames(ptr, 5100000, B, n_signals=50, snr_range=(-10.0, 10.0))
h.enqueue(ptr, B)
got = gather_results_device(h, B, dst=0, force=True)          # D2D -> RCCL all_reduce / all_gather / gather on CUDA tensors -> D2H
want = h.fetch(B)
assert int(want[1].sum()) > 50 * B and int(want[3].max()) > 20
assert got[0].tobytes() == want[0].tobytes() and np.array_equal(got[1], want[1]) and np.array_equal(got[3], want[3])
for f in range(B):
    n = min(int(want[3][f]), _lib.EVENT_CAP)
    assert got[2][f, :n].tobytes() == want[2][f, :n].tobytes(), f
# the packed gather (what bench.py runs inside its timed steps): pack kernels -> device buffer -> RCCL all_gather of the sizes + gather
# on a side stream -> D2H into page-locked memory; `repeat` = rank 0's load of that many ranks.  Byte-identical to the numpy twin
# of the pack kernels applied to fetch()'s arrays, for three batches in a row (both result slots, two gathers in flight).
from pyft8_amd.distributed import PackedGather
g = PackedGather(h, B, dst=0, force=True, repeat=3)
seen = []
for k in range(3):
    h.enqueue(ptr, B)
    res = h.fetch(B)
    g.submit()
    seen.append(_lib.pack_results(*res))
    if k > 0:
        parts = g.collect()
        assert len(parts) == 3 and all(p.buf.tobytes() == seen[k - 1].tobytes() for p in parts), k
parts = g.drain()
assert len(parts) == 3 and all(p.buf.tobytes() == seen[2].tobytes() for p in parts)
pk = parts[0]
assert pk.n_frames == B and 1500 < pk.nbytes / B < 8000, pk.nbytes / B
m_dense = _lib.package_batch(*res)
m_packed = _lib.package_packed(pk)
assert m_dense[0].tobytes() == m_packed[0].tobytes() and np.array_equal(m_dense[1], m_packed[1]) and int(m_dense[1].sum()) > 20 * B
r2, c2, e2, ec2 = pk.expand()
dec = res[0]["status"] == _lib.ST_DECODED
assert np.array_equal(c2, res[1]) and np.array_equal(ec2, res[3]) and r2[dec].tobytes() == res[0][dec].tobytes()      # fetch()'s decoded set, byte for byte
g.close()
# batch sizes that change from call to call (smaller batches reuse the same pack buffers; the receive rows grow when a larger one comes),
# collected either at once or one batch late
rng = np.random.default_rng(3)
PackedGather.MIN_ROW = 4096
g = PackedGather(h, B, dst=0, force=True, repeat=2)
queue = []
for k in range(24):
    n = int(rng.integers(1, 40)) if k < 6 else int(rng.integers(1, B + 1))
    h.enqueue(ptr, n)
    res = h.fetch(n)
    g.submit()
    queue.append(_lib.pack_results(*res))
    if k % 3 == 0 or g.outstanding() > 1:
        parts = g.collect()
        w = queue.pop(0)
        assert len(parts) == 2 and all(p.buf.tobytes() == w.tobytes() for p in parts), k
while queue:
    parts = g.collect()
    w = queue.pop(0)
    assert all(p.buf.tobytes() == w.tobytes() for p in parts)
assert g.outstanding() == 0
g.close()
via_host = gather_results(*want, dst=0, force=True)             # host arrays through the same backend (one H2D per array)
assert via_host[0].tobytes() == want[0].tobytes() and np.array_equal(via_host[1], want[1])
# a zero-copy view of the same batch is the same bytes
h.enqueue(ptr, B)
v = h.fetch_view(B)
assert v[0].tobytes() == want[0].tobytes()
for f in range(B):
    n = min(int(v[3][f]), _lib.EVENT_CAP)
    assert sorted(v[2][f, :n].tolist()) == sorted(want[2][f, :n].tolist()), f        # (log order inside a frame is not deterministic)
dist.barrier(); dist.destroy_process_group(); h.close()
print("RCCL_ONE_RANK_OK")
'''


def test_rccl_device_gather_byte_equal_single_rank_group(tmp_path):
    """A real config-1 batch through ft8rx_results_to_device -> RCCL collectives on CUDA tensors -> D2H, in a one-rank nccl group:
    byte-identical to fetch().  This is the code the 8-GPU run of configs 3/4 executes; it had never run on any machine."""
    script = tmp_path / "nccl_one_rank.py"
    script.write_text(_NCCL_ONE_RANK)
    out = subprocess.run([sys.executable, str(script), ROOT, str(_port())], capture_output=True, text=True, timeout=900, cwd=ROOT,
                         env={k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")})
    assert out.returncode == 0 and "RCCL_ONE_RANK_OK" in out.stdout, (out.stdout + out.stderr)[-3000:]


def test_packed_results_equal_the_numpy_twin_of_fetch():
    """ft8rx_set_packed_output / ft8rx_packed_results: the three pack kernels at the end of a batch write header | frame table | kept
    records | used events -- byte for byte what _lib.pack_results (numpy) makes of the same batch's fetch(); into device memory and
    straight into page-locked host memory; large (event-log compaction, free-running chunks) and small (single chain) batches; an
    undersized buffer is flagged, never overrun."""
    import torch
    from pyft8_amd import _lib
    for B in (320, 6):
        h = _lib.Handle(max_frames=B)
        ptr = h.staging_ptr()
        h.synth_frames(ptr, 5200000, B, n_signals=50, snr_range=(-10.0, 10.0))
        cap = _lib.packed_capacity(B, h.cfg.max_cands)
        dev = [torch.zeros(cap + 64, dtype=torch.uint8, device="cuda") for _ in range(2)]
        h.set_packed_output(dev[0].data_ptr(), dev[1].data_ptr(), cap)
        slots = []
        for k in range(3):
            h.enqueue(ptr, B)
            res = h.fetch(B)
            which, hdr = h.packed_results()
            slots.append(which)
            want = _lib.pack_results(*res)
            assert hdr["bytes"] == len(want) and not hdr["overflow"] and hdr["n_frames"] == B and hdr["max_cands"] == h.cfg.max_cands
            got = dev[which][:hdr["bytes"]].cpu().numpy()
            assert got.tobytes() == want.tobytes(), (B, k)
            assert not dev[which][cap:].any()                               # nothing beyond the capacity
        assert slots[0] != slots[1] and slots[0] == slots[2]                 # the two result slots alternate
        pk = _lib.Packed(got)
        assert int(pk.frames["n_rec"].sum()) == len(pk.records) > 20 * B and pk.nbytes / B < 8000
        # page-locked host buffers: the kernels write across PCIe, no copy
        pin = [h.pinned_bytes(cap) for _ in range(2)]
        h.set_packed_output(pin[0].ctypes.data, pin[1].ctypes.data, cap)
        h.enqueue(ptr, B)
        res = h.fetch(B)
        which, hdr = h.packed_results()
        assert pin[which][:hdr["bytes"]].tobytes() == _lib.pack_results(*res).tobytes()
        m1, m2 = _lib.package_batch(*res), _lib.package_packed(pin[which][:hdr["bytes"]])
        assert m1[0].tobytes() == m2[0].tobytes() and np.array_equal(m1[1], m2[1])
        # the synchronous entry fills it too
        audio = h.download_audio(ptr, B)
        res = h.decode_batch(audio)
        which, hdr = h.packed_results()
        assert pin[which][:hdr["bytes"]].tobytes() == _lib.pack_results(*res).tobytes()
        # too small: header and frame table only, flagged
        small = 32 + 16 * B + 4800
        dev[0].zero_(); dev[1].zero_()
        h.set_packed_output(dev[0].data_ptr(), dev[1].data_ptr(), small)
        h.enqueue(ptr, B)
        res = h.fetch(B)
        which, hdr = h.packed_results()
        assert hdr["overflow"] and hdr["bytes"] == len(_lib.pack_results(*res)) and not dev[which][32 + 16 * B:].any()
        with pytest.raises(_lib.Ft8rxError):
            _lib.Packed(dev[which].cpu().numpy())
        # ordinary (pageable) host memory is refused: the kernels could not write it
        plain = [np.zeros(cap, np.uint8) for _ in range(2)]
        with pytest.raises(_lib.Ft8rxError, match="packed_output"):
            h.set_packed_output(plain[0].ctypes.data, plain[1].ctypes.data, cap)
        # off again
        h.set_packed_output(None, None, 0)
        h.enqueue(ptr, B)
        h.fetch(B)
        with pytest.raises(_lib.Ft8rxError):
            h.packed_results()
        h.close()


def test_packed_results_edge_cases():
    """Packed results for the shapes the config-1 test does not reach: frames with no candidate at all (silence) next to busy ones,
    one-frame batches, max_cands = 256 (every lane of the pack kernels' masks in use), the wide-layout build, and extension knobs that
    log many events per frame -- always byte-identical to the numpy twin of the same batch's fetch(), and rendering the same messages."""
    import torch
    from pyft8_amd import _lib, synth
    busy = np.stack([synth.make_frame(9100 + i, n_signals=60, snr_range=(-6.0, 12.0)) for i in range(3)])
    audio = np.concatenate([np.zeros((2, _lib.NSAMP), np.int16), busy, np.zeros((1, _lib.NSAMP), np.int16)])
    cases = [dict(), dict(max_cands=256, sync_score_min=60.0), dict(f0_hi=1888), dict(osd_triple=20, osd_max_hd=40, bp_iters_b=30)]
    for kw in cases:
        for B in (len(audio), 1):
            h = _lib.Handle(_lib.default_config(**kw), max_frames=B)
            cap = _lib.packed_capacity(B, h.cfg.max_cands)
            dev = [torch.zeros(cap, dtype=torch.uint8, device="cuda") for _ in range(2)]
            h.set_packed_output(dev[0].data_ptr(), dev[1].data_ptr(), cap)
            res = h.decode_batch(audio[2:2 + B] if B == 1 else audio)
            which, hdr = h.packed_results()
            want = _lib.pack_results(*res)
            got = dev[which][:hdr["bytes"]].cpu().numpy()
            assert hdr["bytes"] == len(want) and got.tobytes() == want.tobytes(), (kw, B)
            pk = _lib.Packed(got)
            if B > 1:
                assert int(pk.frames["n_rec"][0]) == 0 and int(pk.frames["n_cand"][0]) == 0 and int(pk.frames["n_rec"][2]) > 20
            m1, m2 = _lib.package_batch(*res), _lib.package_packed(pk)
            assert m1[0].tobytes() == m2[0].tobytes() and np.array_equal(m1[1], m2[1]) and int(m1[1].sum()) >= (10 if B > 1 else 3)
            h.close()


def test_eight_ranks_one_gpu_gloo_uneven_total():
    """What the 8-GPU node will run cold, at world size 8 on the one GPU of a test box (gloo; RCCL refuses several ranks per device):
    bench.py's whole N > 1 flow with an UNEVEN total (8 x 16 + 3 frames: shard() gives the first three ranks one more), per-rank
    placement on pairwise-disjoint whole-core CPU sets, the packed gather inside every timed step with the frames arriving in shard
    order, value = all frames x steps / max-over-ranks time."""
    total, steps = 8 * 16 + 3, 2
    out = _launch(8, ["--backend", "gloo", "--total-frames", str(total), "--steps", str(steps), "--warmup", "1", "--no-host-entry", "--min-seconds", "0"],
                  timeout=2400)
    assert out.returncode == 0, (out.stderr + out.stdout)[-4000:]
    d = _one_line(out)
    assert d["n_gpus"] == 8 and d["steps"] == steps and d["scaling"] == "weak" and d["cpu_baseline"] is None and d["other_configs"] is None
    pr = d["per_rank"]
    assert pr["frames"] == [17, 17, 17, 16, 16, 16, 16, 16]
    assert all(len(pr[k]) == 8 for k in ("ms_per_step", "frames_per_s", "kernel_only_frames_per_s", "placement"))
    # per-rank clocks stop BEFORE the closing barrier (a gloo barrier of 8 processes: milliseconds, against two short steps), `value` after it
    assert d["value"] <= 1.001 * total / (max(pr["ms_per_step"]) * 1e-3) and d["ms_per_step"] >= max(pr["ms_per_step"]) - 1e-3
    assert abs(d["value"] - total / (d["ms_per_step"] * 1e-3)) < 1e-6 * d["value"]
    g = pr["gather"]
    assert g["ok"] and "packed results of 8 ranks gathered to rank 0 over gloo inside every timed step" in d["config"]["gather"], d["config"]["gather"]
    assert "every rank's frames arrived in shard order" in d["config"]["gather"]
    # placement: every rank pinned to its own CPUs; hardware threads of one core never split between two ranks
    sets = []
    for p in pr["placement"]:
        assert p["cpus"], p
        cpus = set()
        for part in p["cpus"].split(","):
            a, _, b = part.partition("-")
            cpus |= set(range(int(a), int(b or a) + 1))
        sets.append(cpus)
    if len(set().union(*sets)) >= 16:                  # (a box with fewer cores than ranks cannot give disjoint slices)
        for i in range(8):
            for j in range(i + 1, 8):
                assert not (sets[i] & sets[j]), (i, j, pr["placement"][i], pr["placement"][j])
        for i, cs in enumerate(sets):
            for c in cs:
                try:
                    sib = open(f"/sys/devices/system/cpu/cpu{c}/topology/thread_siblings_list").read().strip()
                except OSError:
                    continue
                sibs = set()
                for part in sib.split(","):
                    a, _, b = part.partition("-")
                    sibs |= set(range(int(a), int(b or a) + 1))
                assert not any(sibs & sets[j] for j in range(8) if j != i), (i, c, sib)


def test_a_straggler_does_not_slow_the_other_ranks():
    """No cross-rank rendezvous inside the steps: eight ranks over gloo on one GPU, rank 5 sleeps 20 ms on the host
    in every step.  Its own steps take >= 20 ms; every other sending rank keeps its own pace -- each rank only waits for its OWN sends,
    and rank 0's ring (depth >= the number of batches) never fills.  With round 4's per-step all_gather of the byte counts all eight
    clocks were the straggler's.  (Rank 0 is left out: its clock includes draining the last batch of every rank.)"""
    steps, delay = 16, 20.0
    common = ["--backend", "gloo", "--frames", "8", "--steps", str(steps), "--warmup", "2", "--no-host-entry", "--min-seconds", "0",
              "--gather-depth", "32"]
    out = _launch(8, common + ["--delay-rank", "5", "--delay-ms", str(delay)], timeout=2400)
    assert out.returncode == 0, (out.stderr + out.stdout)[-4000:]
    d = _one_line(out)
    ms = d["per_rank"]["ms_per_step"]
    assert d["per_rank"]["gather"]["ok"] and "every rank's frames arrived in shard order" in d["config"]["gather"], d["config"]["gather"]
    others = [ms[r] for r in range(1, 8) if r != 5]
    assert ms[5] >= delay, ms
    assert max(others) < ms[5] - 0.5 * delay, ms                           # they did not inherit the straggler's 20 ms
    ph = d["per_rank"]["gather"]["submit_phases_ms_by_rank"]
    assert all(ph[r]["sizes"] < 1.0 and ph[r]["wait_slot"] < 0.25 * delay for r in range(1, 8)), ph      # no rank waited for another's batch
    assert abs(d["value"] - 8 * 8 / (d["ms_per_step"] * 1e-3)) < 1e-6 * d["value"] and d["ms_per_step"] >= ms[5] - 1e-3


def test_bench_gpus_flag_starts_the_launcher_itself():
    """`python bench.py --gpus 2` with no WORLD_SIZE in the environment = the 2-rank run (bench.py spawns torch.distributed.run as a
    child before touching the GPU); a WORLD_SIZE that contradicts --gpus is an error, not a silently mislabelled run."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo"] + SMALL,
                         capture_output=True, text=True, timeout=1500, cwd=ROOT, env=env)
    assert out.returncode == 0, out.stderr[-3000:]
    assert _one_line(out)["n_gpus"] == 2
    bad = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1"] + SMALL, capture_output=True, text=True,
                         timeout=300, cwd=ROOT, env=dict(env, WORLD_SIZE="2", RANK="0", LOCAL_RANK="0"))
    assert bad.returncode != 0 and "WORLD_SIZE=2" in (bad.stderr + bad.stdout)


def _digest(res, n, cap):
    rec, cnt, ev, evc = res
    h = hashlib.sha256()
    for f in range(n):
        h.update(rec[f, :cnt[f]].tobytes())
        h.update(np.sort(ev[f, :min(int(evc[f]), cap)], order=["cand", "ipass", "slot", "seq"]).tobytes())
    return h.hexdigest()


def test_config3_shard_full_size():
    """BASELINE config 3 = 65 536 frames over 8 GPUs: one rank's contiguous shard of 8192 frames as bench.py --config 3 runs it."""
    from pyft8_amd import _lib
    from pyft8_amd.distributed import shard
    assert shard(65536, 3, 8) == (3 * 8192, 8192)
    B = 8192
    h = _lib.Handle(max_frames=B)
    ptr = h.staging_ptr()
    start, count = shard(65536, 3, 8)
    truth = h.synth_frames(ptr, 7000000 + start, count, n_signals=50, snr_range=(-10.0, 10.0))
    h.set_streams(4)
    h.enqueue(ptr, B)
    big = h.fetch(B)
    small = _lib.Handle(max_frames=32)
    for s0 in (0, 4096, B - 32):                                          # frames decode the same alone as inside the shard
        small.enqueue(ptr + s0 * _lib.NSAMP * 2, 32)
        assert _digest(small.fetch(32), 32, _lib.EVENT_CAP) == _digest(tuple(a[s0:s0 + 32] for a in big), 32, _lib.EVENT_CAP), s0
    small.close()
    msgs, mcnt, flags = _lib.package_batch(*big, return_flags=True)
    assert not flags.any()
    got = sum(len({b" ".join(m["f"]).decode() for m in msgs[f, :mcnt[f]]} & {t["msg"] for t in truth[f]}) for f in range(0, B, 128))
    assert got / (B // 128) > 25
    h.close()
    # and the bench preset names the workload
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--config", "3", "--steps", "1", "--warmup", "1", "--no-host-entry",
                          "--no-cpu-baseline", "--min-seconds", "0"], capture_output=True, text=True, timeout=1500, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-3000:]
    d = _one_line(out)
    assert d["config"]["frames_per_gpu"] == 8192 and "8192 synthetic 15-s frames per GPU" in d["config"]["workload"] and d["other_configs"] is None


def test_config4_low_snr_order3_full_size():
    """BASELINE config 4 per GPU: 2048 frames, <= 10 signals at -24..-20 dB, OSD order 3 (extension) with the distance gate."""
    from pyft8_amd import _lib, messages as M
    B = 2048
    kw = dict(osd_triple=30, osd_max_hd=32)
    cfg = _lib.default_config(**kw)
    h = _lib.Handle(cfg, max_frames=B)
    ptr = h.staging_ptr()
    truth = h.synth_frames(ptr, 8000000, B, n_signals=10, snr_range=(-24.0, -20.0))
    h.enqueue(ptr, B)
    rec, cnt, ev, evc = h.fetch(B)
    msgs, mcnt, flags = _lib.package_batch(rec, cnt, ev, evc, return_flags=True)
    true_hits = false_hits = 0
    for f in range(B):
        want = {t["msg"] for t in truth[f]}
        got = {b" ".join(m["f"]).decode() for m in msgs[f, :mcnt[f]]}
        true_hits += len(got & want)
        false_hits += len(got - want)
    assert not (flags & _lib.PKG_MSG_TRUNCATED).any()
    assert false_hits / B < 0.5                           # the gate keeps order 3 from flooding the list (ungated: several per frame)
    # a sample of frames against the oracle run with the same knobs: identical records and messages
    audio = h.download_audio(ptr, 6)
    ocfg = O.default_config(**_lib.fft_plans(), **kw)
    for f in range(6):
        r = O.decode_frame(audio[f], ocfg)
        assert int(cnt[f]) == len(r["cands"])
        for i, c in enumerate(r["cands"]):
            g = rec[f, i]
            assert (int(g["status"]), int(g["f0_idx"]), int(g["h0_idx"])) == (c.status, c.f0_idx, c.h0_idx)
            if c.status == 1:
                assert (int(g["ipass"]), int(g["ap"]), int(g["method"]), int(g["n_its"]), int(g["msg_lo"]), int(g["msg_hi"])) == \
                       (c.ipass, c.ap, c.method, c.n_its, c.msg_lo, c.msg_hi)
        mine = [" ".join(m["msg_tuple"]) for m in M.package_frame(rec[f], int(cnt[f]), ev[f], int(evc[f]))]
        assert mine == [" ".join(m["msg_tuple"]) for m in r["msgs"]]
    h.close()
    print(f"config 4 sample: {true_hits / B:.3f} true / {false_hits / B:.3f} false decodes per frame at -24..-20 dB")
