"""Randomised GPU-vs-oracle parity sweep (GPU box): many frames with random recipes and Receiver kwargs / decoder knobs, every
candidate record and every rendered message compared with the CPU oracle (oracle/ = the checker; nothing here is product code).
Usage: python tools/parity_sweep.py [n_batches] [frames_per_batch] [seed] [first_frame_index]  -> one line per batch, a summary at the end"""
import os
import sys
import time
from concurrent.futures import ProcessPoolExecutor

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
from pyft8_amd import _lib, messages as M, synth  # noqa: E402
from pyft8_amd.receiver import config_from_kwargs  # noqa: E402

KW = [dict(), dict(), dict(sync_score_min=100, max_cands=150), dict(search_freq_range=[300, 2500], search_time_range=[-1.0, 2.0]),
      dict(search_freq_range=[100, 5900]), dict(search_freq_range=[1500, 4200], max_cands=90), dict(bp_iters_b=30, osd_single=40, osd_double=4),
      dict(osd_triple=20, osd_max_hd=34), dict(max_cands=256, sync_score_min=70),
      dict(osd_single=91, osd_double=3),
      # round 6: wide time windows (symbols read clamped, among them the one that starts exactly at the clamp position) and more
      # candidates than the default layouts hold (max_cands > 256: the deep layouts of libft8rx_wide.so)
      dict(search_time_range=[-1.0, 8.2], sync_score_min=70), dict(search_time_range=[2.0, 8.0], sync_score_min=60, max_cands=256),
      dict(search_time_range=[-6.0, 3.0], sync_score_min=70), dict(max_cands=600, sync_score_min=40),
      # ... and windows beyond the fine-sync series (k_fine_td, the sync search in several launches), up to the reference's own limits
      dict(search_time_range=[-20.0, 20.0], sync_score_min=70), dict(search_time_range=[-36.4, 22.6], sync_score_min=80)]
KNOBS = ("bp_nc0_a", "bp_iters_a", "bp_nc0_b", "bp_iters_b", "osd_single", "osd_double", "osd_triple", "osd_max_hd", "llr_sd_min")


def oracle_frame(args):
    import oracle as O
    audio, c = args
    ocfg = O.default_config(**c)
    r = O.decode_frame(audio, ocfg)
    recs = [(x.f0_idx, x.h0_idx, np.float32(x.score).tobytes(), x.status, (x.ipass, x.ap, x.method, x.n_its, x.msg_lo, x.msg_hi) if x.status == 1 else None,
             (x.ttweak, x.ftweak, x.nsync) if x.status in (1, 4, 5) and (x.status != 1 or x.ipass >= 2) else None) for x in r["cands"]]
    return recs, [" ".join(m["msg_tuple"]) for m in r["msgs"]]


def run_sweep(nb=20, fpb=16, seed=20260102, kw_list=None, first_index=9000000, verbose=True):
    """nb batches of fpb random frames, batch b decoded with kw_list[b % len(kw_list)] (default: the ten sets of KW) and random
    stream counts / ladder modes; -> dict(frames, cands, msgs, bad, seconds, kwargs_sets).  Used by the command line below and by the
    driver-run test tests/test_gpu_parity.py::test_randomised_parity_sweep."""
    kw_list = KW if kw_list is None else kw_list
    rng = np.random.default_rng(seed)
    tot = dict(frames=0, cands=0, msgs=0, bad=0)
    t0 = time.time()
    with ProcessPoolExecutor(max_workers=min(32, os.cpu_count() or 8)) as pool:
        for b in range(nb):
            kw = kw_list[b % len(kw_list)]
            cfg = config_from_kwargs(**kw)
            wide = cfg.f0_hi > 960
            frames = []
            for k in range(fpb):
                ns = int(rng.choice([0, 1, 5, 20, 50, 70]))
                lo = float(rng.choice([-24.0, -18.0, -10.0, 0.0]))
                frames.append(synth.make_frame(first_index + 1000 * b + k, n_signals=ns, snr_range=(lo, lo + float(rng.choice([6.0, 14.0, 25.0]))),
                                               freq_range=(150.0, 5650.0) if wide else (200.0, 2800.0)))
            audio = np.stack(frames)
            h = _lib.Handle(cfg, max_frames=fpb)
            h.set_streams(int(rng.choice([1, 2, 4])))
            h.set_ladder_mode(int(rng.integers(0, 2)))
            rec, cnt, ev, evc = h.decode_batch(audio)
            msgs, mcnt = _lib.package_batch(rec, cnt, ev, evc)
            h.close()
            oc = dict(sync_score_min=cfg.sync_score_min, max_cands=cfg.max_cands, f0_lo=cfg.f0_lo, f0_hi=cfg.f0_hi, h0_lo=cfg.h0_lo, h0_hi=cfg.h0_hi)
            oc.update({k: getattr(cfg, k) for k in KNOBS})
            bad = 0
            for f, (orecs, otxt) in enumerate(pool.map(oracle_frame, [(audio[f], oc) for f in range(fpb)])):
                n = int(cnt[f])
                grecs = [(int(r["f0_idx"]), int(r["h0_idx"]), np.float32(r["score"]).tobytes(), int(r["status"]),
                          (int(r["ipass"]), int(r["ap"]), int(r["method"]), int(r["n_its"]), int(r["msg_lo"]), int(r["msg_hi"])) if int(r["status"]) == 1 else None,
                          (int(r["ttweak"]), int(r["ftweak"]), int(r["nsync"])) if int(r["status"]) in (1, 4, 5) and (int(r["status"]) != 1 or int(r["ipass"]) >= 2) else None)
                         for r in rec[f, :n]]
                gtxt = [b" ".join(m["f"]).decode() for m in msgs[f, :mcnt[f]]]
                ptxt = [" ".join(m["msg_tuple"]) for m in M.package_frame(rec[f], n, ev[f], int(evc[f]))]
                ok = grecs == orecs and gtxt == otxt and ptxt == otxt
                bad += not ok
                tot["cands"] += n; tot["msgs"] += len(otxt)
            tot["frames"] += fpb; tot["bad"] += bad
            if verbose:
                print(f"batch {b:3d} kwargs {kw}: {fpb} frames, {'IDENTICAL' if not bad else f'{bad} FRAMES DIFFER'}", flush=True)
    tot["seconds"] = time.time() - t0
    tot["kwargs_sets"] = min(nb, len(kw_list))
    return tot


def main():
    nb = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    fpb = int(sys.argv[2]) if len(sys.argv) > 2 else 16
    seed = int(sys.argv[3]) if len(sys.argv) > 3 else 20260102              # other frames and recipes: another seed / first frame index
    first = int(sys.argv[4]) if len(sys.argv) > 4 else 9000000
    tot = run_sweep(nb, fpb, seed=seed, first_index=first)
    print(f"{tot['frames']} frames, {tot['cands']} candidate records, {tot['msgs']} messages: {tot['bad']} frames differ from the oracle "
          f"({tot['seconds']:.0f} s)")
    sys.exit(1 if tot["bad"] else 0)


if __name__ == "__main__":
    main()
