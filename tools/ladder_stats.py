import sys, numpy as np
sys.path.insert(0, '.')
from pyft8_amd import _lib
h = _lib.Handle(_lib.default_config(), max_frames=256)
ptr = h.staging_ptr()
h.synth_frames(ptr, 12345, 256, n_signals=50, snr_range=(-10.0, 10.0))
h.enqueue(ptr, 256)
rec, cnt, ev, evc = h.fetch(256)
st = []; 
tot = 0
from collections import Counter
c = Counter()
for f in range(256):
    r = rec[f, :cnt[f]]
    tot += cnt[f]
    for s, ip, ap in zip(r["status"], r["ipass"], r["ap"]):
        c[(int(s), int(ip) if s == 1 else -1, int(ap) if s == 1 else -1)] += 1
print("candidates", tot)
for k in sorted(c): print(k, c[k])
