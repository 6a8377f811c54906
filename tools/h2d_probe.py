"""The pipelined host entry alone (ft8rx_enqueue_batch_host: H2D of batch k+1 behind the kernels of batch k), for a rocprofv3 trace
(GPU box):  python tools/h2d_probe.py [frames] [steps] [mode]   mode: host (default) | device | sync"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pyft8_amd import _lib  # noqa: E402


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 30
    mode = sys.argv[3] if len(sys.argv) > 3 else "host"
    h = _lib.Handle(max_frames=B)
    ptr = h.staging_ptr()
    h.synth_frames(ptr, 0, B)
    host = h.download_audio(ptr, B)
    pinned = [h.pinned_audio(B), h.pinned_audio(B)]
    for p in pinned:
        p[:] = host
    dev = _lib.Handle(max_frames=B)                 # a second handle whose staging buffer keeps the frames resident on the device
    dev.decode_batch(host)
    d_ptr = dev.staging_ptr()
    nt = min(32, len(os.sched_getaffinity(0)))

    def loop(n):
        for i in range(n):
            if mode == "host":
                h.enqueue_host(pinned[i & 1])
            elif mode == "device":
                h.enqueue(d_ptr, B)
            else:
                _lib.package_batch(*h.decode_batch(pinned[0]), n_threads=nt)
                continue
            if i > 0:
                _lib.package_batch(*h.fetch_view(B), n_threads=nt)
        if mode != "sync":
            _lib.package_batch(*h.fetch_view(B), n_threads=nt)
        h.sync()
    loop(4)
    t0 = time.perf_counter()
    loop(steps)
    dt = time.perf_counter() - t0
    print(f"{mode}: {B * steps / dt:.0f} frames/s, {1e3 * dt / steps:.3f} ms/step")


if __name__ == "__main__":
    main()
