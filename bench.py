"""bench.py -- FT8 15-s frames decoded per second on N MI355X (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W [--frames B | --config 1|2|3|4]

Launched as one process per GPU (`python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N`); when --gpus N > 1
is given WITHOUT a torchrun environment, bench.py starts that launcher itself as a child process (before anything touches the GPU)
and exits with its code, so `python bench.py --gpus 8` is the 8-GPU run and never a mislabelled single-GPU one.

One "step" = one pass of the whole receive hot path (spectrogram -> Costas sync -> LLR -> cycle FFT ->
fine sync -> LDPC BP -> OSD -> records -> D2H -> host message layer: every message tuple rendered) over one
batch of B synthetic 15-s frames per GPU (BASELINE config 1: 50 signals/frame, -10..+10 dB).  The audio is
resident in HBM before the timed region; the host work of batch k-1 overlaps the GPU work of batch k.
Frames are independent, so ranks shard them with no collective on the decode path (weak scaling).  With more
than one rank (or --force-gather) every batch's PACKED results -- the decoded / event-logging candidates' records
and the used event log, ~4 KB per frame -- are gathered to rank 0 over RCCL INSIDE the timed steps, on a side
stream that overlaps the next batch (pyft8_amd/distributed.py: PackedGather); `value` includes it.
At N = 1 the default command then also runs BASELINE configs 2, 3 (one rank's shard) and 4 (per GPU) for a few steps
each and nests them under "other_configs" (outside `value`).
Rank 0 prints ONE JSON line (see DESIGN.md section "Measurement" for the roofline accounting).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

# multi-process GPU work on this pool needs dmabuf IPC (the host driver has no legacy IPC: RCCL fails with hipIpcGetMemHandle otherwise);
# the image exports it, a bare environment may not -- set before anything loads the HIP runtime
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# algorithmic HBM bytes per frame of each kernel group (DESIGN.md / SURVEY.md 8d), N_c = 200 candidates
ALG_BYTES = {
    "spectrogram": 360000 + 375 * 976 * 4,                 # int16 audio in + dB grid out
    "sync": 375 * 976 * 4,                                 # grid read
    "topk": 928 * 8,
    "grid_llr": 200 * 58 * 8 * 4 + 200 * 174 * 4,          # payload gather + llr out
    "bp_grid": 200 * 174 * 4,
    "select0": 200 * 48,
    "cycle_fft": 360000 + 47414 * 8,                       # audio in + the spectrum bins ever read
    "fine": 200 * 1064 * 8 + 200 * 174 * 4,                # spectrum slices + llr out
    "bp_fine": 200 * 174 * 4 * 2,
    "select1": 200 * 48,
    "osd": 200 * 174 * 4,
    "select2": 200 * 48,
}
ALG_BYTES_FRAME = 6025712                                  # SURVEY.md 8d total
HBM_PEAK_GBS = 8000.0                                      # MI355X_MICROARCH.md: 8 TB/s spec
# algorithmic fp32 operations per frame (reference-shaped dataflow: spectrogram 43 M + sync 23 M + cycle FFT 8 M +
# fine 0.78 G + BP 0.43 G worst case + OSD ~0.03 G) against the fp32 vector peak of MI355X_MICROARCH.md (157.3 TFLOP/s with FMA)
ALG_FLOP_FRAME = 1.3e9
ALG_FLOP_FINE = 0.78e9
VALU_PEAK_TFLOPS = 157.3
# executed fp32 operations of k_fine per candidate (counted from the kernel; an fma counts 2): the pruned IFFT of the
# time scan 140 k (radix-8 pass with its zero inputs left out 29 k + [4,4] stage 54 k + [5,5] stage with the last pass pruned 57 k) + its 56
# symbol DFTs on lane quads (0.6 k each); EIGHT FREQUENCY-DOMAIN SCORES since round 4 (per residue 10 complex multiplies and 70
# complex-by-real multiply-adds, x 100 residues = 34 k; then per tone 26 items of four residues, 16 adds + 6 symbols x 4 fmas each, x 7 tones
# = 12 k, + the 16-lane sums 1.5 k: 48 k each, against 140 k + 7 x 0.6 k for the pruned IFFT + symbol DFTs they replace); THE FINAL GRID
# also from the slice (H for eight tones 38 k, 160 ten-point transforms + twiddles + magnitudes 26 k, and for the 30 % of candidates with
# clamped symbols one more H pass: 76 k on average, against 160 k + 45 x 0.6 k for the full IFFT + symbol DFTs)
EXEC_FLOP_FINE_CAND = 140e3 + 56 * 0.6e3 + 8 * 48e3 + 76e3
# rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this command (tools/pmc_summary.py, collected by tools/collect_profiles.sh),
# by BASELINE configuration (the exact command `bench.py --config N`; config 1 = the default command)
PMC_PROFILES = {1: os.path.join(ROOT, "profiles", "pmc_latest.json"), 2: os.path.join(ROOT, "profiles", "pmc_config2_latest.json"),
                3: os.path.join(ROOT, "profiles", "pmc_config3_latest.json"), 4: os.path.join(ROOT, "profiles", "pmc_config4_latest.json")}


SQ_PROFILE = os.path.join(ROOT, "profiles", "sq_latest.json")      # rocprofv3 --pmc SQ_* passes of `bench.py --streams 1 --subbatch 0` (tools/pmc_sq.sh)
# vector-instruction issue ceiling: 256 CUs x 4 SIMDs, one wave64 VALU instruction per 2 cycles per SIMD (a SIMD executes 32 lanes of a
# plain fp32 / int32 operation per cycle: tools/ubench/valu_rate.hip, profiles/archive/r02_valu_rate.txt), 2.4 GHz
VALU_ISSUE_PEAK = 1024 * 2.4e9 / 2
STAGE_KERNELS = {"spectrogram": ["k_spectrogram"], "sync": ["k_sync"], "topk": ["k_topk"], "grid_llr": ["k_grid_llr", "k_worklist_att"], "bp_grid": ["k_bp"],
                 "select0": ["k_select0"], "cycle_fft": ["k_cyc_a", "k_cyc_bc"], "fine": ["k_fine", "k_worklist"], "bp_fine": ["k_bp", "k_select1"],
                 "select1": ["k_select1"], "osd": ["k_osd", "k_osd_nan"], "select2": ["k_select2"]}


def source_hash():
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    from src_hash import source_hash as sh
    return sh()


def profile_stale(path):
    """True when a stored counter profile was collected from other native sources than the tree being benchmarked (None: no stamp)."""
    try:
        stamp = json.load(open(path)).get("_source_hash")
    except Exception:
        return None
    return None if stamp is None else stamp != source_hash()


def roofline_valu(acc, B, dom):
    """The roofline that describes this path (VALU / latency bound, not HBM bound): executed vector instructions (wave64
    instructions, SQ_INSTS_VALU of the committed SQ pass, per 256-frame launch) against the issue ceiling, for the dominant kernel and for
    the whole step.  k_bp runs four times per step (one grid-stage and three fine-stage launches); the profile holds its per-launch average."""
    if B != 256 or not os.path.exists(SQ_PROFILE):
        return None
    prof = json.load(open(SQ_PROFILE))
    launches = {"k_bp": 4, "k_select1": 3}
    insts = {k: v.get("INSTS_VALU", 0.0) * launches.get(k, 1) for k, v in prof.items() if isinstance(v, dict)}
    step_ms = sum(acc.values())
    dk = STAGE_KERNELS.get(dom, [dom])[0]
    if dk not in insts:
        return None
    whole = sum(v for k, v in insts.items() if k.startswith("k_") and k not in ("k_synth", "k_fill_row0"))
    return {"unit": "wave64 VALU instructions", "issue_peak_per_s": VALU_ISSUE_PEAK, "kernel": dk,
            "insts_per_launch": insts[dk], "kernel_ms": acc[dom], "frac": insts[dk] / (acc[dom] * 1e-3) / VALU_ISSUE_PEAK,
            "step_insts": whole, "step_ms_sum_of_stages": step_ms, "step_frac": whole / (step_ms * 1e-3) / VALU_ISSUE_PEAK,
            "source": "profiles/sq_latest.json (rocprofv3 --pmc SQ_INSTS_VALU pass of `bench.py --streams 1 --subbatch 0`, tools/pmc_sq.sh; not measured in this run)",
            "stale": profile_stale(SQ_PROFILE)}


def gpu_use_reading():
    """One mid-run reading of the driver's own utilisation counter, to stderr (so that a coarse external sampler that reports 0 can be
    told from an idle GPU): rocm-smi --showuse, or amd-smi metric -u; silent when neither tool is there."""
    import shutil
    import subprocess
    for cmd in (["rocm-smi", "--showuse"], ["amd-smi", "metric", "-u"]):
        exe = shutil.which(cmd[0]) or (os.path.join("/opt/rocm/bin", cmd[0]) if os.path.exists(os.path.join("/opt/rocm/bin", cmd[0])) else None)
        if not exe:
            continue
        try:
            out = subprocess.run([exe] + cmd[1:], capture_output=True, text=True, timeout=20).stdout
            lines = [ln.strip() for ln in out.splitlines() if "use" in ln.lower() or "GFX" in ln or "busy" in ln.lower()]
            print(f"[bench.py] {' '.join(cmd)} during the timed loop: " + " | ".join(lines[:6]), file=sys.stderr)
            return
        except Exception as e:
            print(f"[bench.py] {' '.join(cmd)} failed: {type(e).__name__}: {e}", file=sys.stderr)


def pmc_traffic(kernel_stage, B):
    """HBM bytes per launch of the stage's kernels from the committed PMC profile of this configuration (B = its key in PMC_PROFILES, None
    for a workload no profile was collected for), or None."""
    names = {"spectrogram": ["k_spectrogram"], "sync": ["k_sync"], "fine": ["k_fine"], "osd": ["k_osd"],
             "cycle_fft": ["k_cyc_a", "k_cyc_b", "k_cyc_c"], "grid_llr": ["k_grid_llr"], "topk": ["k_topk"]}.get(kernel_stage)
    if not names or B not in PMC_PROFILES or not os.path.exists(PMC_PROFILES[B]):
        return None
    prof = json.load(open(PMC_PROFILES[B]))
    if not all(n in prof for n in names):
        return None
    return sum(prof[n]["hbm_bytes"] for n in names)


def per_kernel_hbm(acc, B):
    """Measured HBM GB/s of every stage that has PMC traffic on file for this configuration: bytes per launch / HIP-event duration."""
    if B not in PMC_PROFILES or not os.path.exists(PMC_PROFILES[B]):
        return None
    prof = json.load(open(PMC_PROFILES[B]))
    groups = {"spectrogram": ["k_spectrogram"], "sync": ["k_sync"], "grid_llr": ["k_grid_llr"], "cycle_fft": ["k_cyc_a", "k_cyc_b", "k_cyc_c"],
              "fine": ["k_fine"], "osd": ["k_osd"]}
    out = {}
    for stage, names in groups.items():
        if stage in acc and all(n in prof for n in names):
            gbs = sum(prof[n]["hbm_bytes"] for n in names) / (acc[stage] * 1e-3) / 1e9
            out[stage] = {"GB/s": round(gbs, 1), "frac_of_peak": round(gbs / HBM_PEAK_GBS, 4)}
    return out


def _gen(args):
    from pyft8_amd import synth
    i, nsig, lo, hi = args
    return synth.make_frame(i, n_signals=nsig, snr_range=(lo, hi))


def make_frames(start, count, nsig=50, snr=(-10.0, 10.0)):
    import multiprocessing as mp
    n = min(count, max(1, min(32, (os.cpu_count() or 1))))
    with mp.get_context("fork").Pool(n) as pool:
        frames = pool.map(_gen, [(start + i, nsig, snr[0], snr[1]) for i in range(count)], chunksize=max(1, count // (4 * n)))
    return np.stack(frames)


def _oracle_worker(args):
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import oracle as O
    frames, plans = args
    cfg = O.default_config(**plans)
    for f in frames:
        O.decode_frame(f, cfg)
    return len(frames)


def cpu_all_cores(frames, plans):
    """The same oracle on all host cores (one process per core, the sample frames repeated): frames/s and the core count."""
    import multiprocessing as mp
    cores = min(os.cpu_count() or 1, 64)
    per = 4
    jobs = [(frames[[(i * per + j) % len(frames) for j in range(per)]], plans) for i in range(cores)]
    with mp.get_context("spawn").Pool(cores) as pool:
        pool.map(_oracle_worker, [(frames[:1], plans)] * cores)          # start the workers, load the library
        t0 = time.perf_counter()
        n = sum(pool.map(_oracle_worker, jobs))
        dt = time.perf_counter() - t0
    return {"value": n / dt, "unit": "frames/s", "cores": cores, "sample": f"{n} frame decodes over {cores} processes, {dt:.1f} s"}


def cpu_baseline(frames, budget_s=15.0, gpu_texts=None):
    """The CPU oracle (single thread) on a bounded sample of the same frames.  gpu_texts (the GPU's message texts per frame, emit order, for
    the same frames at the same -- reference -- knobs): the metric's "decode-set match vs CPU ref" on the frames the baseline decodes anyway."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import oracle as O
    from pyft8_amd import _lib
    cfg = O.default_config(**_lib.fft_plans())
    O.decode_frame(frames[0], cfg)
    t0 = time.perf_counter()
    n = 0
    texts = []
    while n < len(frames) and time.perf_counter() - t0 < budget_s:
        texts.append([" ".join(m["msg_tuple"]) for m in O.decode_frame(frames[n], cfg)["msgs"]])
        n += 1
    dt = time.perf_counter() - t0
    out = {"value": n / dt, "unit": "frames/s", "cores": 1, "kind": "port",
           "sample": f"first {n} frames of this rank-0 batch, oracle/ft8_oracle.c single-threaded, {dt:.1f} s"}
    if gpu_texts is not None:
        same = sum(1 for i in range(n) if i < len(gpu_texts) and gpu_texts[i] == texts[i])
        out["decode_set_match"] = {"frames_compared": n, "frames_identical": same, "messages": sum(len(t) for t in texts),
                                   "what": "message texts in emit order, GPU (last timed batch) vs this CPU oracle, frame by frame"}
    try:
        out["all_cores"] = cpu_all_cores(frames, _lib.fft_plans())
    except Exception as e:                          # informational: never lose the line over it
        out["all_cores"] = {"error": f"{type(e).__name__}: {e}"}
    return out


def _cpulist(text):
    out = []
    for part in text.strip().split(","):
        if part:
            a, _, b = part.partition("-")
            out += list(range(int(a), int(b or a) + 1))
    return out


def _fmt_cpus(cpus):
    """[0,1,2,3,8,9] -> "0-3,8-9"."""
    out, i = [], 0
    cpus = sorted(cpus)
    while i < len(cpus):
        j = i
        while j + 1 < len(cpus) and cpus[j + 1] == cpus[j] + 1:
            j += 1
        out.append(f"{cpus[i]}-{cpus[j]}" if j > i else f"{cpus[i]}")
        i = j + 1
    return ",".join(out)


def _split_whole_cores(cpus, k, n):
    """Part k of n of a CPU list, keeping the hardware threads of one core together (so that two ranks never share a physical core)."""
    groups, seen = [], set()
    for c in sorted(cpus):
        if c in seen:
            continue
        try:
            sib = [x for x in _cpulist(open(f"/sys/devices/system/cpu/cpu{c}/topology/thread_siblings_list").read()) if x in cpus]
        except OSError:
            sib = [c]
        sib = sib or [c]
        seen.update(sib)
        groups.append(sorted(sib))
    per = max(1, len(groups) // n)
    mine = groups[k * per:(k + 1) * per] or groups
    return sorted(c for g in mine for c in g)


def place_rank(local_rank, world, bus_id_of):
    """Pin this rank (and every thread it starts later: the packaging pool, RCCL's and HIP's helper threads) to its own slice of host
    cores -- on the NUMA node of its GPU when sysfs says which one that is -- so that eight ranks do not float over both sockets and
    page-locked buffers allocated afterwards are first-touched next to the GPU that reads them.  bus_id_of(r) -> PCI address of rank
    r's GPU (or None).  -> a dict describing the placement (it goes into the JSON line).  Never fatal."""
    info = {"cpus": None, "numa_node": None, "how": "unpinned"}
    try:
        allowed = sorted(os.sched_getaffinity(0))

        def node_of(r):
            bdf = bus_id_of(r)
            path = f"/sys/bus/pci/devices/{bdf.lower()}/numa_node" if bdf else None
            return int(open(path).read().strip()) if path and os.path.exists(path) else None
        nodes = [node_of(r) for r in range(world)]
        node = nodes[local_rank]
        info["gpu_pci"] = bus_id_of(local_rank)
        node_cpus = None
        lst = f"/sys/devices/system/node/node{node}/cpulist" if node is not None and node >= 0 else None
        if lst and os.path.exists(lst):
            node_cpus = [c for c in _cpulist(open(lst).read()) if c in allowed]
        if node_cpus:
            peers = [r for r in range(world) if nodes[r] == node]          # ranks whose GPUs hang off the same node share its cores
            k, n = peers.index(local_rank), len(peers)
            mine = _split_whole_cores(node_cpus, k, n)
            info.update(numa_node=node, how=f"slice {k + 1}/{n} of NUMA node {node} (whole cores)")
        else:
            mine = _split_whole_cores(allowed, local_rank, max(1, world))
            info.update(how=f"slice {local_rank + 1}/{world} of the {len(allowed)} allowed CPUs (whole cores; no NUMA node known for the GPU)")
        os.sched_setaffinity(0, mine)
        info["cpus"] = _fmt_cpus(mine)
        info["n_cpus"] = len(mine)
    except Exception as e:                                   # placement is an optimisation
        info["how"] = f"unpinned ({type(e).__name__}: {e})"
    return info


OTHER_CONFIGS = {      # BASELINE.json configs 2-4 per GPU, as `--config N` runs them; steps chosen so that the un-overlapped last fetch weighs < 2 %
    2: dict(frames=4096, signals=50, snr=(-10.0, 10.0), knobs=dict(bp_iters_b=30, osd_single=30, osd_double=2), steps=14,
            what="4096 frames, LDPC BP 30 iterations + OSD depth 2"),
    3: dict(frames=8192, signals=50, snr=(-10.0, 10.0), knobs={}, steps=10, what="one rank's 8192-frame shard of the 65 536-frame job, Receiver defaults"),
    4: dict(frames=2048, signals=10, snr=(-24.0, -20.0), knobs=dict(osd_triple=30, osd_max_hd=32), steps=24,
            what="2048 frames per GPU, 10 signals at -24..-20 dB, OSD order 3 over 30 positions with the distance gate at 32"),
}


def run_other_config(n, device, rank, pk_threads, streams, subbatch=None):
    """A short run of BASELINE config n on this GPU, measured like the headline: pipelined steps to rendered messages, then per-stage
    HIP-event times of whole-batch launches.  -> the dict nested under other_configs[n]."""
    import torch
    from pyft8_amd import _lib
    c = OTHER_CONFIGS[n]
    B = c["frames"]
    cfg = _lib.default_config(**c["knobs"])
    t_setup = time.perf_counter()
    h = _lib.Handle(cfg=cfg, device=device, max_frames=B)
    try:
        h.set_streams(streams)
        if subbatch is not None:
            h.set_subbatch(subbatch)
        d_audio = torch.empty((B, _lib.NSAMP), dtype=torch.int16, device="cuda")
        truth = h.synth_frames(d_audio.data_ptr(), rank * 1000000, B, n_signals=c["signals"], snr_range=c["snr"])
        setup_s = time.perf_counter() - t_setup

        def steps(k):
            for i in range(k):
                h.enqueue(d_audio.data_ptr(), B)
                if i > 0:
                    _lib.package_batch(*h.fetch_view(B), n_threads=pk_threads)
            return _lib.package_batch(*h.fetch_view(B), n_threads=pk_threads)
        steps(1)
        h.sync()
        t0 = time.perf_counter()
        msgs, mcnt = steps(c["steps"])
        h.sync()
        dt = time.perf_counter() - t0
        t1 = time.perf_counter()                     # kernels only (no D2H wait, no host layer), pipelined over the chunk streams like the timed loop
        for _ in range(c["steps"]):
            h.enqueue(d_audio.data_ptr(), B)
        h.sync()
        kernel_only = B * c["steps"] / (time.perf_counter() - t1)
        h.set_profiling(True)
        samples = {}
        for _ in range(3):
            h.enqueue(d_audio.data_ptr(), B)
            h.sync()
            for k, v in h.stage_times().items():
                samples.setdefault(k, []).append(v)
        h.set_profiling(False)
        acc = {k: float(np.median(v)) for k, v in samples.items()}
        dom = max(acc, key=acc.get)
        achieved = ALG_BYTES[dom] * B / (acc[dom] * 1e-3) / 1e9
        # what the rate decodes: messages of the last timed batch against the generator's truth (a rate without a yield says nothing
        # about decoding -- config 4's frames are below the reference algorithm's reach and its rate is the front end's)
        n_true = n_false = 0
        for f in range(B):
            want = {t["msg"] for t in truth[f]}
            got = {" ".join(x.decode() for x in msgs[f, i]["f"] if x) for i in range(int(mcnt[f]))}
            n_true += len(got & want)
            n_false += len(got - want)
        return {"workload": c["what"], "value": B * c["steps"] / dt, "unit": "frames/s", "ms_per_step": 1e3 * dt / c["steps"], "steps": c["steps"],
                "frames_per_gpu": B, "messages_per_frame": float(mcnt.mean()),
                "true_decodes_per_frame": n_true / B, "false_decodes_per_frame": n_false / B, "signals_per_frame": c["signals"],
                "decoded_fraction_of_signals": (n_true / B / c["signals"]) if c["signals"] else None,
                "stage_ms": {k: round(v, 4) for k, v in acc.items()},
                "kernels_only_frames_per_s": kernel_only, "stage_sum_frames_per_s": B / (sum(acc.values()) * 1e-3),
                "roofline": {"kernel": dom, "kernel_ms": acc[dom], "achieved": achieved, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                             "traffic": pmc_traffic(dom, n), "whole_path_frac": B * c["steps"] / dt * ALG_BYTES_FRAME / 1e9 / HBM_PEAK_GBS},
                "setup_s": round(setup_s, 2)}
    finally:
        h.close()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100, help="timed steps (default 100 = 0.5 s of config 1: the un-overlapped fetch + host layer of the LAST step, 1.3 ms, then weighs 0.3 % instead of 1.3 % at 20)")
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--frames", type=int, default=256, help="frames per GPU per step (BASELINE config 1: 256)")
    ap.add_argument("--total-frames", type=int, default=None, help="N > 1: shard THIS many frames per step over the ranks (contiguous blocks, "
                    "pyft8_amd.distributed.shard: the first total %% N ranks get one more) instead of --frames per GPU -- an uneven split for flow tests")
    ap.add_argument("--unique", type=int, default=64, help="(host generator only) distinct frames generated per rank, tiled to --frames")
    ap.add_argument("--host-synth", action="store_true", help="generate frames with the numpy generator instead of the device kernel")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-host-entry", action="store_true", help="skip the informational host-pointer (PCIe-inclusive) passes, whose "
                    "chunked launches would mix part-batch kernels into a rocprofv3 per-kernel average")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend for N > 1 (nccl = RCCL; gloo only for flow tests on one GPU)")
    ap.add_argument("--force-gather", action="store_true", help="run the gather collectives even with one rank (a one-rank process group is "
                    "initialised): the RCCL device path on a one-GPU box")
    ap.add_argument("--no-gather", action="store_true", help="N > 1: leave the per-step gather of the packed results to rank 0 out (A/B of its cost)")
    ap.add_argument("--gather-repeat", type=int, default=1, help="measurement aid: gather every batch this many times (rank 0's receive + D2H load "
                    "of that many ranks in a one-rank group)")
    ap.add_argument("--gather-depth", type=int, default=4, help="receive sets rank 0 keeps per peer (PackedGather depth): how many batches a rank may run "
                    "ahead of the slowest one before its sends wait")
    ap.add_argument("--delay-rank", type=int, default=None, help="test aid: this rank sleeps --delay-ms on the host in every step (a straggler)")
    ap.add_argument("--delay-ms", type=float, default=20.0)
    ap.add_argument("--render-gathered", action="store_true", help="rank 0 also renders the messages of ALL gathered frames inside the timed steps "
                    "(by default every rank renders its own shard and rank 0 keeps the packed form of the others)")
    ap.add_argument("--no-other-configs", action="store_true", help="N = 1 default command: skip the short runs of BASELINE configs 2-4")
    ap.add_argument("--other-configs", action="store_true", help="run the short config 2-4 passes even for a non-default workload")
    ap.add_argument("--host-threads", type=int, default=None, help="threads of the native host message layer per rank (default: min(32, this rank's CPUs))")
    ap.add_argument("--no-pin", action="store_true", help="leave the rank's CPU affinity alone")
    ap.add_argument("--share-gpu", action="store_true", help="flow tests on a box with fewer GPUs than ranks: rank r uses device r %% device_count "
                    "(gloo always does this; RCCL itself refuses two ranks on one device)")
    ap.add_argument("--signals", type=int, default=50, help="signals per synthetic frame (config 1/2: 50, config 4: <= 10)")
    ap.add_argument("--snr", type=float, nargs=2, default=(-10.0, 10.0), metavar=("LO", "HI"), help="SNR range in dB / 2500 Hz")
    ap.add_argument("--bp-iters", type=int, default=None, help="extension knob: iterations of the second BP stage (reference 20; config 2: 30)")
    ap.add_argument("--osd", type=int, nargs=2, default=None, metavar=("SINGLE", "DOUBLE"), help="extension knob: osd_012 flip counts (reference 30 2)")
    ap.add_argument("--streams", type=int, default=2, help="HIP streams a batch is cut across (1 = one chain of whole-batch launches)")
    ap.add_argument("--subbatch", type=int, default=None, help="frames per kernel chain inside a stream's share of a batch (ft8rx_set_subbatch; "
                    "default: the library's 256; 0 = the whole share in one chain, the round-4 behaviour)")
    ap.add_argument("--config", type=int, default=None, choices=(1, 2, 3, 4), help="BASELINE.json configuration preset (per GPU): 1 = 256 frames; "
                    "2 = 4096 frames, BP 30 iterations, OSD depth 2; 3 = 8192 frames per GPU (65 536 over 8 GPUs) + record gather; "
                    "4 = 2048 frames per GPU (16 384 over 8), <= 10 signals at -24..-20 dB, OSD order 3 over 30 positions with the distance gate at 32")
    ap.add_argument("--osd3", type=int, default=None, help="extension knob: OSD order-3 depth (triple flips over the first N basis positions; 0 = off)")
    ap.add_argument("--osd-max-hd", type=int, default=None, help="extension knob: accept an OSD trial only within this Hamming distance of the hard decisions (0 = off)")
    ap.add_argument("--queue-depth", type=int, default=2, choices=(1, 2), help="batches queued on the GPU behind the one that computes: 2 = "
                    "results are copied out of their slot and the slot is re-enqueued before the batch is packaged (absorbs host-side delays "
                    "of up to a step); 1 = the round-4 loop (views of the result slot, one batch queued)")
    ap.add_argument("--min-seconds", type=float, default=3.0, help="after the K timed steps keep stepping (untimed for `value`, reported as "
                    "extra_steps) until the GPU has been busy this long, so that coarse GPU-activity samplers see the run")
    args = ap.parse_args()
    explicit = {a.split("=")[0] for a in sys.argv[1:] if a.startswith("--")}
    if args.config is not None:
        preset = {1: dict(frames=256), 2: dict(frames=4096, bp_iters=30, osd=(30, 2)), 3: dict(frames=8192),
                  4: dict(frames=2048, signals=10, snr=(-24.0, -20.0), osd3=30, osd_max_hd=32)}[args.config]
        for k, v in preset.items():
            if "--" + k.replace("_", "-") not in explicit:
                setattr(args, k, v)
        if args.config >= 2 and "--steps" not in explicit:
            args.steps, args.warmup = 3, 1

    world_env = os.environ.get("WORLD_SIZE")
    if world_env is None and args.gpus > 1:
        # not under a launcher: start one as a CHILD process (nothing has touched the GPU yet) and pass its exit code on
        import socket
        import subprocess
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
               "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        raise SystemExit(subprocess.call(cmd))
    if int(world_env or 1) != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world_env or 1}: launch one process per GPU "
                         f"(python -m torch.distributed.run --nproc-per-node {args.gpus} bench.py --gpus {args.gpus} ...)")

    # (GPU_MAX_HW_QUEUES is deliberately left alone: eight hardware queues would give the gather's side stream a queue of its own, but
    # they cost the decode kernels 9 % -- 53.9 k -> 49.1 k frames/s kernels only, profiles/archive/r04_notes.md.  With the default four the gather's
    # copies may sit behind a chunk's kernel chain for up to one batch; nothing waits for them, the byte counts travel over gloo.)
    import torch
    import torch.distributed as dist
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the product has no CPU path")
    if args.backend != "nccl" or args.share_gpu:
        local = local % torch.cuda.device_count()          # flow test: several ranks may share one GPU
    torch.cuda.set_device(local)
    # CPU/NUMA placement before anything allocates page-locked memory or starts threads (the handle, the packaging pool)
    from pyft8_amd import _lib, messages
    placement = {"how": "unpinned (--no-pin)"}
    orig_affinity = os.sched_getaffinity(0)
    if not args.no_pin:
        ndev = max(1, torch.cuda.device_count())
        placement = place_rank(int(os.environ.get("LOCAL_RANK", "0")), world, lambda r: _lib.device_pci_bus_id(r % ndev))
    if world == 1 and args.force_gather:
        import datetime
        import socket
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        kw = dict(device_id=torch.device("cuda", local)) if args.backend == "nccl" else {}
        dist.init_process_group(args.backend, init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1,
                                timeout=datetime.timedelta(seconds=300), **kw)
    if world > 1:
        import datetime
        tmo = datetime.timedelta(seconds=300)      # a rank that dies must not leave the others in a collective for the default 10+ minutes
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local), timeout=tmo)
        else:
            dist.init_process_group(args.backend, timeout=tmo)
    B = args.frames
    shard_counts, frame0 = [args.frames] * world, rank * 1000000          # frames per rank and step; index of this rank's first synthetic frame
    if args.total_frames is not None:
        from pyft8_amd.distributed import shard
        shard_counts = [shard(args.total_frames, r, world)[1] for r in range(world)]
        frame0, B = shard(args.total_frames, rank, world)
        if min(shard_counts) < 8:
            raise SystemExit(f"bench.py: --total-frames {args.total_frames} leaves a rank fewer than 8 frames")
    job_frames = sum(shard_counts)                                           # frames of the whole job per step
    # which committed PMC profile describes this workload: the BASELINE configuration asked for, or config 1 for the plain default command
    default_workload = (args.frames, args.signals, tuple(args.snr), args.bp_iters, args.osd, args.osd3, args.osd_max_hd) == (256, 50, (-10.0, 10.0), None, None, None, None)
    knob_flags = {"--frames", "--signals", "--snr", "--bp-iters", "--osd", "--osd3", "--osd-max-hd"}
    pmc_key = (args.config if not (knob_flags & explicit) else None) if args.config is not None else (1 if default_workload else None)
    cfg = _lib.default_config()
    if args.bp_iters is not None:
        cfg.bp_iters_b = args.bp_iters
    if args.osd is not None:
        cfg.osd_single, cfg.osd_double = args.osd
    if args.osd3 is not None:
        cfg.osd_triple = args.osd3
    if args.osd_max_hd is not None:
        cfg.osd_max_hd = args.osd_max_hd
    knobs = (f"BP {cfg.bp_iters_a}/{cfg.bp_iters_b} iters, OSD {cfg.osd_single}/{cfg.osd_double}" + (f"/order-3 over {cfg.osd_triple}" if cfg.osd_triple else "")
             + (f", distance gate {cfg.osd_max_hd}" if cfg.osd_max_hd else ""))
    reference_knobs = (cfg.bp_iters_b, cfg.osd_single, cfg.osd_double, cfg.osd_triple, cfg.osd_max_hd) == (20, 30, 2, 0, 0)
    h = _lib.Handle(cfg=cfg, device=local, max_frames=B)
    h.set_streams(args.streams)
    if args.subbatch is not None:
        h.set_subbatch(args.subbatch)
    if args.host_synth:
        uniq = min(args.unique, B)
        frames = make_frames(frame0, uniq, args.signals, tuple(args.snr))
        reps = (B + uniq - 1) // uniq
        d_audio = torch.from_numpy(np.concatenate([frames] * reps)[:B]).cuda()
        data_desc = f"{uniq} distinct numpy-generated frames tiled to {B}"
    else:
        # B distinct frames per rank, generated on the GPU (k_synth: GFSK + Philox noise), config-1 recipe
        uniq = B
        d_audio = torch.empty((B, _lib.NSAMP), dtype=torch.int16, device="cuda")
        h.synth_frames(d_audio.data_ptr(), frame0, B, n_signals=args.signals, snr_range=tuple(args.snr))
        frames = d_audio[:min(B, 192)].cpu().numpy()  # CPU-baseline sample (bounded by cpu_baseline's 15-s budget)
        data_desc = f"{B} distinct device-generated frames"
    torch.cuda.synchronize()

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # One step = one batch through the whole path: kernels on the GPU, results to the host (double-buffered D2H on a copy stream)
    # and the native host message layer (every message tuple rendered), pipelined: while batch k computes, the host fetches and
    # packages batch k-1.  All K batches are fully decoded to message arrays inside the timed region.
    cores = len(os.sched_getaffinity(0))                   # this rank's slice after place_rank
    pk_threads = max(2, min(64, cores if not args.no_pin else cores // max(1, world)))      # (256 frames: 0.56 ms at 32 threads, 0.41 at 64)
    if args.host_threads:
        pk_threads = max(1, args.host_threads)

    # The gather (BASELINE config 3 "RCCL gather of decoded messages"): with more than one rank every batch's packed results go to
    # rank 0 inside the step -- submit() after the fetch of batch k-1, while batch k computes; the last one is drained before the
    # clock stops.  Every rank renders the messages of its own shard (the host layer scales with the ranks); rank 0 holds the
    # packed form of all shards and can render any frame from it (--render-gathered does, for all of them, inside the step).
    gather = None
    gather_disabled = None
    if (world > 1 or args.force_gather) and not args.no_gather:
        from pyft8_amd.distributed import PackedGather
        # set the gather up and push ONE batch through it before anything is timed; if that fails on any rank (a transport missing on
        # this box, ...) every rank drops the gather -- agreed through one all_reduce -- and the run is still measured, with the reason
        # in the line, instead of dying inside the timed loop
        err = None
        try:
            gather = PackedGather(h, max(shard_counts), dst=0, force=args.force_gather, repeat=args.gather_repeat, depth=args.gather_depth)
            h.enqueue(d_audio.data_ptr(), B)
            h.fetch_view(B)
            gather.submit()
            gather.drain()
        except Exception as e:
            err = f"{type(e).__name__}: {e}"
        ok = torch.tensor([0 if err else 1], device="cuda" if args.backend == "nccl" else "cpu", dtype=torch.int32)
        if dist.is_initialized() and world > 1:
            dist.all_reduce(ok, op=dist.ReduceOp.MIN)
        if int(ok.item()) == 0:
            gather_disabled = err or "the gather failed on another rank"
            try:
                if gather is not None:
                    h.set_packed_output(None, None, 0)
            except Exception:
                pass
            gather = None
    rendered = [0]

    def gather_side():
        if args.delay_rank is not None and rank == args.delay_rank:
            time.sleep(args.delay_ms * 1e-3)
        if gather is not None:
            gather.submit()
            if args.render_gathered and rank == 0:
                parts = gather.collect() if gather.outstanding() > 1 else None      # the batch before: its gather has landed
                for pk in parts or []:
                    rendered[0] += int(_lib.package_packed(pk, n_threads=pk_threads)[1].sum())

    def host_side(view):
        gather_side()
        return _lib.package_batch(*view, n_threads=pk_threads)

    reuse = {}                          # result / message arrays of the two-deep loop, allocated by its first step
    step_marks = []                     # host clock after every step's host side (the timed region fills it: where a slow run lost its time)

    def enq_resident(i):
        h.enqueue(d_audio.data_ptr(), B)

    def run_steps(n, marks=None, enq=enq_resident):
        if args.queue_depth < 2:
            # one batch queued behind the one that computes: while batch i computes the host fetches and packages batch i - 1
            # (views of the handle's page-locked result buffers, no copy)
            for i in range(n):
                enq(i)
                if i > 0:
                    host_side(h.fetch_view(B))
                    if marks is not None:
                        marks.append(time.perf_counter())
            out = host_side(h.fetch_view(B))
        else:
            # two batches queued: the results of batch i are COPIED out of their slot (ft8rx_fetch_results), its gather is started, and
            # batch i + 2 is enqueued into that slot before batch i is packaged -- the GPU always has a whole batch waiting behind the
            # one that computes, so a host-side delay of up to one step (a loaded host: the packaging threads or this thread lose
            # their CPUs for milliseconds) no longer idles it
            out = None
            for i in range(min(2, n)):
                enq(i)
            for i in range(n):
                if i + 2 < n:
                    res = h.fetch(B, out=reuse.get("res"))    # (into the same arrays every step: no 10 MB of allocation, first-touch
                    reuse["res"] = res                         #  faults and unmapping per 3-ms step under 64 threads)
                else:
                    res = h.fetch_view(B)                      # the last two batches: nothing is enqueued into their slots any more
                gather_side()               # (before the enqueue: the pack kernels of batch i + 2 wait for this send through the fence)
                if i + 2 < n:
                    enq(i + 2)
                reuse["msg"] = _lib.package_batch(*res, n_threads=pk_threads, return_flags=True, out=reuse.get("msg"))
                out = reuse["msg"][:2]
                if marks is not None and i + 1 < n:
                    marks.append(time.perf_counter())
        if gather is not None:
            gather.drain()                  # the last batch's results have reached rank 0's host memory
        return out

    if args.warmup:
        run_steps(args.warmup)
    h.sync()
    # the interpreter's cyclic collector stays out of the timed region (a full collection in the middle of a 60-ms region is a
    # double-digit percentage of it): collect now, switch it off, back on afterwards
    import gc
    gc.collect()
    gc.disable()
    barrier()
    t0 = time.perf_counter()
    msgs_, mc_ = run_steps(args.steps, step_marks)
    h.sync()
    dt_own = time.perf_counter() - t0          # this rank's own K steps (its gathers drained), before it meets the others
    barrier()
    dt = time.perf_counter() - t0
    gc.enable()
    step_gaps = np.diff(np.array([t0] + step_marks)) * 1e3 if step_marks else np.zeros(1)
    # keep the GPU busy for --min-seconds in total (identical steps, not part of `value`): a 0.16-s timed region is invisible to
    # a GPU-activity sampler with a period of seconds
    extra_steps, extra_dt = 0, 0.0
    if args.min_seconds > dt:
        n_more = max(1, int((args.min_seconds - dt) / (dt / args.steps)))
        t_e = time.perf_counter()
        smi = None
        if rank == 0:                                    # one utilisation reading while the same loop keeps the GPU busy (outside `value`)
            import threading
            smi = threading.Thread(target=gpu_use_reading, daemon=True)
            smi.start()
        run_steps(n_more)
        h.sync()
        extra_steps, extra_dt = n_more, time.perf_counter() - t_e
        if smi is not None:
            smi.join(timeout=30)
    # kernel-only rate (no D2H, no host layer), informational
    h.sync()
    t1 = time.perf_counter()
    for _ in range(args.steps):
        h.enqueue(d_audio.data_ptr(), B)
    h.sync()
    kernel_only = B * args.steps / (time.perf_counter() - t1)
    cdev = "cuda" if args.backend == "nccl" else "cpu"
    t = torch.tensor([dt], device=cdev, dtype=torch.float64)
    own = torch.tensor([dt_own, kernel_only], device=cdev, dtype=torch.float64)   # this rank's own clock (before the closing barrier), for the per-rank table
    per_rank_raw = [own]
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        per_rank_raw = [torch.zeros_like(own) for _ in range(world)]
        dist.all_gather(per_rank_raw, own)
    dt = float(t.item())
    per_rank_raw = [[float(x) for x in r.tolist()] for r in per_rank_raw]

    # per-kernel timing (HIP events on the library's stream), outside the timed region
    h.set_profiling(True)
    samples = {}
    for _ in range(5):
        h.enqueue(d_audio.data_ptr(), B)
        h.sync()
        for k, v in h.stage_times().items():
            samples.setdefault(k, []).append(v)
    acc = {k: float(np.median(v)) for k, v in samples.items()}       # median of 5 launches per stage
    h.set_profiling(False)
    rec, cnt, ev, evc = h.fetch(B)
    # informational: the host-pointer entry (ft8rx_decode_batch: pageable host audio -> H2D -> kernels -> D2H), PCIe inclusive
    pcie = pcie_pinned = pcie_sync = None
    if not args.no_host_entry:
        host_audio = d_audio.cpu().numpy()
        h.decode_batch(host_audio)
        t2 = time.perf_counter()
        for _ in range(3):
            h.decode_batch(host_audio)
        pcie = 3 * B / (time.perf_counter() - t2)                       # synchronous entry, pageable memory
        pinned = [h.pinned_audio(B), h.pinned_audio(B)]                 # page-locked host memory (ft8rx_alloc_host), two input buffers
        for pb in pinned:
            pb[:] = host_audio
        h.decode_batch(pinned[0])
        t2 = time.perf_counter()
        for _ in range(3):
            h.decode_batch(pinned[0])
        pcie_sync = 3 * B / (time.perf_counter() - t2)                  # synchronous entry, pinned memory
        # the metric as SURVEY 8d defines it (H2D + kernels + D2H + host message layer): the SAME loop as the timed region above (two
        # batches queued, every message rendered, the same --steps, the cyclic collector off), only the audio comes from page-locked
        # HOST memory -- batch k + 2's audio crosses PCIe while batches k, k + 1 compute (ft8rx_enqueue_batch_host); the gather is
        # not part of this pass (this rank only)
        keep_gather, gather = gather, None
        try:
            run_steps(2, enq=lambda i: h.enqueue_host(pinned[i & 1]))
            h.sync()
            gc.collect()
            gc.disable()
            t2 = time.perf_counter()
            run_steps(args.steps, enq=lambda i: h.enqueue_host(pinned[i & 1]))
            h.sync()
            pcie_pinned = args.steps * B / (time.perf_counter() - t2)
        finally:
            gc.enable()
            gather = keep_gather
        del host_audio, pinned
    n_dec = int(sum((rec[f][:cnt[f]]["status"] == 1).sum() for f in range(B)))
    # candidates that ran the fine-sync kernel: everything not decoded / stopped on the grid LLRs (ipass 0)
    valid = np.arange(rec.shape[1])[None, :] < cnt[:, None]
    n_fine = int((valid & ~(((rec["status"] == 1) & (rec["ipass"] < 2)) | (rec["status"] == 2))).sum())
    n_msgs = sum(len(messages.package_frame(rec[f], int(cnt[f]), ev[f], int(evc[f]))) for f in range(min(B, 16)))

    # the gather ran inside the timed steps; here (outside) rank 0 checks what arrived: one packed part per rank, rank 0's own part
    # against its local results (decoded set byte for byte, messages rendered from the packed form = from the dense arrays)
    gather_note = "single rank: nothing to gather" if not args.no_gather else "gather left out (--no-gather)"
    if gather_disabled:
        gather_note = f"GATHER DISABLED, value is without it: {gather_disabled}"
    gather_info = None
    if gather is not None:
        try:
            h.enqueue(d_audio.data_ptr(), B)
            h.sync()
            h.fetch_view(B)                           # (the un-timed passes above left older batches unfetched)
            view = h.fetch_view(B)                    # the batch just enqueued; the GPU is idle now
            t3 = time.perf_counter()
            gather.submit()
            parts = gather.drain()
            sync_ms = 1e3 * (time.perf_counter() - t3)       # one un-overlapped gather, submit to landed
            # frame identity across the job: every rank's per-frame digest of its decoded records, against the gathered parts in rank order
            import zlib

            def digests(recs_of_frame, n):
                out = []
                for f in range(n):
                    r = recs_of_frame(f)
                    d = r[r["status"] == _lib.ST_DECODED]
                    cols = np.stack([d[k].astype(np.uint64) for k in ("msg_lo", "msg_hi", "f0_idx", "h0_idx", "ipass")]) if len(d) else np.zeros(0, np.uint64)
                    out.append(zlib.crc32(np.ascontiguousarray(cols).tobytes()))
                return out
            mine = digests(lambda f: view[0][f, :view[1][f]], B)
            every = [mine]
            if world > 1:
                every = [None] * world
                dist.all_gather_object(every, mine)
            sub = np.array(gather.seconds[args.warmup + 1:] or gather.seconds)
            gather_info = {"submit_ms_per_step": round(1e3 * float(sub.mean()), 4), "submit_ms_max": round(1e3 * float(sub.max()), 4),
                           "unoverlapped_ms": round(sync_ms, 3), "repeat": args.gather_repeat, "rendered_on_rank0": bool(args.render_gathered),
                           "depth": gather.depth,
                           "submit_phases_ms": {k: round(1e3 * float(np.mean([p.get(k, 0.0) for p in gather.phases[args.warmup + 1:] or gather.phases])), 4)
                                                for k in ("header", "wait_slot", "sizes", "issue")}}
            if world > 1:                             # every rank's own host time in submit(), by phase: no rank waits for another's batch
                allph = [None] * world
                dist.all_gather_object(allph, gather_info["submit_phases_ms"])
                gather_info["submit_phases_ms_by_rank"] = allph
            if rank == 0:
                assert len(parts) == world * args.gather_repeat, f"{len(parts)} parts from {world} ranks"
                total = sum(p.n_frames for p in parts[:world])
                assert [p.n_frames for p in parts[:world]] == shard_counts, f"gathered {[p.n_frames for p in parts[:world]]} frames, expected {shard_counts}"
                order_ok = [digests(lambda f, p=p: p.frame(f)[0], p.n_frames) for p in parts[:world]] == every
                own = parts[0]
                gather_info["bytes_per_frame"] = round(sum(p.nbytes for p in parts[:world]) / total, 1)
                gather_info["records_per_frame"] = round(sum(len(p.records) for p in parts[:world]) / total, 2)
                r2, c2, e2, ec2 = own.expand()
                keep = np.zeros(view[0].shape, bool)
                for f in range(B):
                    keep[f, own.frame(f)[0]["pad2"]] = True
                dec = view[0]["status"] == _lib.ST_DECODED
                ok = (np.array_equal(c2, view[1]) and np.array_equal(ec2, view[3]) and not (dec & ~keep).any()
                      and r2[keep].tobytes() == view[0][keep].tobytes())
                m_dense = _lib.package_batch(*view, n_threads=pk_threads)
                m_packed = _lib.package_packed(own, n_threads=pk_threads)
                ok = ok and m_dense[0].tobytes() == m_packed[0].tobytes() and np.array_equal(m_dense[1], m_packed[1])
                gather_note = (f"packed results of {world} rank{'s' if world > 1 else ' (forced)'} gathered to rank 0 over {args.backend} inside every "
                               f"timed step ({gather_info['bytes_per_frame']:.0f} B/frame; host time in submit {gather_info['submit_ms_per_step']:.3f} ms/step, "
                               f"one un-overlapped gather {sync_ms:.2f} ms); rank 0's own part "
                               + ("= its local decoded set and messages" if ok else "DIFFERS from its local results")
                               + ("; every rank's frames arrived in shard order" if order_ok else "; FRAME ORDER / CONTENT MISMATCH against the ranks' own digests"))
                gather_info["ok"] = bool(ok and order_ok)
        except Exception as e:                    # validation outside the timed region: report, do not lose the line
            gather_note = f"gather check failed: {type(e).__name__}: {e}"

    # the other single-GPU BASELINE configurations, a few steps each, outside `value` (N = 1, default command)
    other = None
    if world == 1 and not args.no_other_configs and ((default_workload and args.config is None) or args.other_configs):
        other = {}
        if gather is not None:
            gather.close()
            gather = None
        h.close()                                   # HBM and page-locked memory back before the 8192-frame handle
        del d_audio
        torch.cuda.empty_cache()
        for n in (2, 3, 4):
            try:
                other[str(n)] = run_other_config(n, local, rank, pk_threads, args.streams, args.subbatch)
            except Exception as e:                    # informational: never lose the line over it
                other[str(n)] = {"error": f"{type(e).__name__}: {e}"}

    placements = [placement]
    if world > 1:
        placements = [None] * world
        dist.all_gather_object(placements, placement)
    if rank == 0:
        dom = max(acc, key=acc.get)
        dom_ms = acc[dom]
        achieved = ALG_BYTES[dom] * B / (dom_ms * 1e-3) / 1e9
        value = job_frames * args.steps / dt
        line = {
            "metric": "FT8 15-s frames decoded/sec", "value": value, "unit": "frames/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * dt / args.steps,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            # What `value` is: K steps of the whole path (kernels, D2H, every message rendered on the host) with the batch's audio RESIDENT
            # IN HBM and the SAME batch decoded in every step (the bench contract's definition).  SURVEY 8d defines the metric with the
            # H2D in it: `value_8d_host_audio` is the same loop over the same K steps with the audio handed over in page-locked HOST
            # memory (this rank only, PCIe inclusive; measured right after the timed region, not part of `value`)
            "value_definition": "audio resident in HBM, same batch every step; kernels + D2H + host message layer inside the timed steps",
            "value_8d_host_audio": pcie_pinned, "value_8d_steps": args.steps if pcie_pinned is not None else None,
            "value_incl_h2d": pcie_pinned, "value_incl_h2d_sync_call": pcie_sync, "value_incl_h2d_sync_call_pageable": pcie,
            "extra_steps": extra_steps, "extra_steps_frames_per_s": (job_frames * extra_steps / extra_dt) if extra_steps else None,
            # host-side gaps between consecutive steps of the timed region on rank 0 (the first one includes the pipeline fill)
            "step_gap_ms": {"p50": float(np.median(step_gaps)), "p90": float(np.percentile(step_gaps, 90)), "max": float(step_gaps.max()),
                            "argmax": int(step_gaps.argmax()), "over_2x_median": int((step_gaps > 2 * np.median(step_gaps)).sum())},
            "config": {"workload": f"{'config 1: ' if (B, args.signals, reference_knobs) == (256, 50, True) else ''}batch of {B} synthetic 15-s frames "
                                   f"per GPU ({data_desc}), {args.signals} signals/frame, {args.snr[0]:+.0f}..{args.snr[1]:+.0f} dB SNR, "
                                   f"{'Receiver defaults' if reference_knobs else 'extension knobs'} ({knobs}); audio resident in HBM, same batch every step",
                       "frames_per_gpu": B, "decoded_candidates_per_frame": n_dec / B,
                       "unique_messages_first16": n_msgs, "messages_per_frame": float(mc_.mean()),
                       "kernel_only_frames_per_s_this_rank": kernel_only, "host_message_threads": pk_threads, "queue_depth": args.queue_depth,
                       "GPU_MAX_HW_QUEUES": os.environ.get("GPU_MAX_HW_QUEUES"),
                       "host_pointer_sync_entry_frames_per_s_pageable": pcie, "host_pointer_sync_entry_frames_per_s_pinned": pcie_sync,
                       "host_pointer_pipelined_entry_frames_per_s_pinned": pcie_pinned, "parallelism": f"frames sharded over {world} GPU(s), no collective on the decode path", "gather": gather_note},
            "roofline": {"bound": "hbm", "kernel": dom, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": pmc_traffic(dom, pmc_key),
                         # the kernel's REAL memory rate: counter traffic / duration / peak (frac above prices the algorithmic bytes of SURVEY
                         # 8d, which the kernel mostly finds in L2: nobody should read it as achieved bandwidth)
                         "counter_frac": (pmc_traffic(dom, pmc_key) / (dom_ms * 1e-3) / 1e9 / HBM_PEAK_GBS) if pmc_traffic(dom, pmc_key) is not None else None,
                         "north_star_note": "the north star's 40 % of HBM roofline = 531 k frames/s of this arithmetic (6.03 MB algorithmic bytes per frame): "
                                            "the path is instruction-issue bound (roofline_valu), not memory bound",
                         "traffic_stale": profile_stale(PMC_PROFILES[pmc_key]) if pmc_key in PMC_PROFILES else None,
                         "traffic_source": f"profiles/{os.path.basename(PMC_PROFILES[pmc_key])} (rocprofv3 --pmc FETCH_SIZE x2 / WRITE_SIZE passes of "
                                           f"`bench.py --config {pmc_key}` at B = {B}, collected by tools/collect_profiles.sh; not measured in this run)" if pmc_traffic(dom, pmc_key) is not None else None,
                         "kernel_ms": dom_ms, "alg_bytes_per_launch": ALG_BYTES[dom] * B,
                         "whole_path_frac": value / world * ALG_BYTES_FRAME / 1e9 / HBM_PEAK_GBS,
                         # measured HBM rate of each stage (PMC bytes / event time): the memory-bound stages sit near the roofline
                         "per_stage_measured_hbm": per_kernel_hbm(acc, pmc_key),
                         # secondary figure SURVEY.md 8d asks for: the path is VALU/LDS/latency bound, not HBM bound
                         "valu": None if (args.signals, tuple(args.snr)) != (50, (-10.0, 10.0)) else {"unit": "TFLOP/s", "peak": VALU_PEAK_TFLOPS,
                                  "whole_path_achieved": value / world * ALG_FLOP_FRAME / 1e12,
                                  "whole_path_frac": value / world * ALG_FLOP_FRAME / 1e12 / VALU_PEAK_TFLOPS,
                                  "fine_achieved": ALG_FLOP_FINE * B / (acc["fine"] * 1e-3) / 1e12,
                                  "fine_frac": ALG_FLOP_FINE * B / (acc["fine"] * 1e-3) / 1e12 / VALU_PEAK_TFLOPS,
                                  # what k_fine really executes: EXEC_FLOP_FINE_CAND fp32 operations per candidate that reaches it (one pruned
                                  # IFFT + frequency-domain scores and grid; an fma counts 2)
                                  "fine_executed_flop_per_launch": EXEC_FLOP_FINE_CAND * n_fine,
                                  "fine_executed_achieved": EXEC_FLOP_FINE_CAND * n_fine / (acc["fine"] * 1e-3) / 1e12,
                                  "fine_executed_frac_of_nofma_peak": EXEC_FLOP_FINE_CAND * n_fine / (acc["fine"] * 1e-3) / 1e12 / (VALU_PEAK_TFLOPS / 2),
                                  "fine_executed_frac_of_fma_peak": EXEC_FLOP_FINE_CAND * n_fine / (acc["fine"] * 1e-3) / 1e12 / VALU_PEAK_TFLOPS,
                                  "fine_candidates_per_launch": n_fine,
                                  "note": "algorithmic fp32 flops of the reference-shaped dataflow (SURVEY 8d: 1.3 GFLOP/frame, "
                                          "fine sync 0.78 G of 18 full IFFTs); the kernel executes ONE pruned IFFT, eight frequency-domain scores and a frequency-domain grid "
                                          "per candidate (0.63 MFLOP), fused multiply-adds only where the arithmetic contract names them (the "
                                          "frequency-domain scores are almost all fma: its ceiling lies between the plain-op and the fma rate)"}},
            "roofline_valu": roofline_valu(acc, B, dom) if default_workload else None,
            "stage_ms": {k: round(v, 4) for k, v in acc.items()},
            "other_configs": other,
            # each rank's own clock over the K timed steps (value uses the max), its kernels-only rate, and where it ran
            "per_rank": {"ms_per_step": [round(1e3 * r[0] / args.steps, 4) for r in per_rank_raw],
                         "frames_per_s": [round(n * args.steps / r[0], 1) for n, r in zip(shard_counts, per_rank_raw)], "frames": shard_counts,
                         "kernel_only_frames_per_s": [round(r[1], 1) for r in per_rank_raw],
                         "placement": placements, "gather": gather_info},
        }
        if not args.no_cpu_baseline:
            # the CPU oracle timed on this box's host cores: rank 0 at N = 1 only (the contract); null in multi-GPU runs
            os.sched_setaffinity(0, orig_affinity)      # the CPU baseline may use every host core, not this rank's slice
            gpu_texts = None
            if world == 1 and reference_knobs and not args.host_synth:           # the same frames, the reference's knobs on both sides
                gpu_texts = [[" ".join(x.decode() for x in msgs_[f, i]["f"]) for i in range(int(mc_[f]))] for f in range(min(B, len(frames)))]      # (all three fields: an empty third one keeps its separator, as " ".join(msg_tuple) does)
            line["cpu_baseline"] = cpu_baseline(frames, gpu_texts=gpu_texts) if world == 1 else None
        try:                                  # whatever native libraries hold in their stdio buffers (RCCL's version banner) goes out first:
            import ctypes                     # the JSON line is the LAST line of stdout
            ctypes.CDLL(None).fflush(None)
        except Exception:
            pass
        print(json.dumps(line), flush=True)
    if gather is not None:
        gather.close()
    if dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()
    h.close()


if __name__ == "__main__":
    main()
