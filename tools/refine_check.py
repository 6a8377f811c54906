import sys, numpy as np
sys.path.insert(0, '.')
from pyft8_amd import _lib, synth
from pyft8_amd.receiver import Receiver
n = 12
rx = Receiver("", None, max_frames=n)
h = rx._handle(n)
truth = h.synth_frames(h.staging_ptr(), 8200000, n, n_signals=50, snr_range=(-10.0, 10.0))
audio = h.download_audio(h.staging_ptr(), n)
want = [{t["msg"]: t for t in truth[f]} for f in range(n)]
one = rx.decode_frames(audio)
for mode in (1, 2):
    h.decode_batch(audio)
    sigs, tr = [], []
    for f in range(n):
        keep = [d for d in one[f] if " ".join(d["msg_tuple"]) in want[f] and int(d["their_snr"]) > -10]
        tr.append([want[f][" ".join(d["msg_tuple"])] for d in keep])
        sigs.append([(synth.tones79(synth.pack77(*d["msg_tuple"])), d["fHz"], d["tsec"]) for d in keep])
    res, orig = h.subtract(h.staging_ptr(), n, sigs, refine=mode, return_origins=True, return_float=True)
    dt = np.array([o[1] - t["t0"] for f in range(n) for o, t in zip(orig[f], tr[f])])
    df = np.array([o[0] - t["f0"] for f in range(n) for o, t in zip(orig[f], tr[f])])
    print("mode", mode, "signals", len(dt), "time error ms: max %.2f rms %.2f" % (1e3*np.abs(dt).max(), 1e3*dt.std()), " freq error Hz: max %.3f rms %.3f" % (np.abs(df).max(), df.std()),
          " residual rms", float(res.std()), "input rms", float(audio.astype(np.float32).std()))
