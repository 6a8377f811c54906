import sys, time
sys.path.insert(0, '.')
from pyft8_amd import _lib
for name, kw in (("default build, search 100-3000 Hz", {}), ("wide build, search 100-5900 Hz", dict(f0_hi=1888))):
    h = _lib.Handle(_lib.default_config(**kw), max_frames=256)
    ptr = h.staging_ptr()
    h.synth_frames(ptr, 5000, 256, n_signals=50, snr_range=(-10.0, 10.0))
    for _ in range(3):
        h.enqueue(ptr, 256)
    h.sync()
    t0 = time.perf_counter()
    for _ in range(20):
        h.enqueue(ptr, 256)
    h.sync()
    dt = (time.perf_counter() - t0) / 20
    h.set_profiling(True); h.enqueue(ptr, 256); h.sync()
    st = h.stage_times()
    print(f"{name}: {256 / dt:.0f} frames/s kernels only; stages ms:", {k: round(v, 3) for k, v in st.items() if v > 0.05})
    h.close()
