"""Why is a big handle slower when it is not the first thing a process does?  kernel-only rate of the config-3 shard (8192 frames)
(a) in a fresh process, (b) after a 256-frame handle has been created, used and closed, (c) after torch has allocated and freed 20 GB."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pyft8_amd import _lib

def rate(B=8192, steps=4):
    h = _lib.Handle(max_frames=B)
    d = torch.empty((B, _lib.NSAMP), dtype=torch.int16, device="cuda")
    h.synth_frames(d.data_ptr(), 0, B)
    h.enqueue(d.data_ptr(), B); h.sync()
    t = time.perf_counter()
    for _ in range(steps):
        h.enqueue(d.data_ptr(), B)
    h.sync()
    r = B * steps / (time.perf_counter() - t)
    h.close(); del d; torch.cuda.empty_cache()
    return r

mode = sys.argv[1]
if mode == "b":
    h = _lib.Handle(max_frames=256); d = torch.empty((256, _lib.NSAMP), dtype=torch.int16, device="cuda")
    h.synth_frames(d.data_ptr(), 0, 256)
    for _ in range(50): h.enqueue(d.data_ptr(), 256)
    h.sync(); h.close(); del d; torch.cuda.empty_cache()
if mode == "c":
    x = [torch.empty(1 << 30, dtype=torch.uint8, device="cuda") for _ in range(20)]; del x; torch.cuda.empty_cache()
print(mode, round(rate()), round(rate()))
