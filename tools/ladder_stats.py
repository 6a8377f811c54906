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
    for s, ip, ap, me in zip(r["status"], r["ipass"], r["ap"], r["method"]):
        c[(int(s), int(ip) if s == 1 else -1, int(ap) if s == 1 else -1, int(me) if s == 1 else -1)] += 1
print("candidates", tot, " key = (status, ipass, ap, method): status 1 decoded / 2 stop grid sd / 3 stop Costas / 4 stop fine sd / 5 exhausted; method 0 GOOD91, 1 LDPC_A, 2 LDPC_B, 3 OSD, 4 LDPC_B+OSD")
for k in sorted(c): print(k, c[k])
