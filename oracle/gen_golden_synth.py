"""Golden of the reference transmitter's waveform: runs PyFT8/transmitter.py:52-70
`symbols_to_complex_audio` in this container (reference imported read-only through ref_harness) on two tone sequences and stores
a sample of the complex waveform -- both amplitude ramps in full plus every 16th sample -- in tests/golden/tx_waveform.npz.
Build container only (needs /root/reference); the committed .npz travels, the reference does not."""
import importlib
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from ref_harness import load_reference  # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    load_reference()
    tx = importlib.import_module("PyFT8.transmitter")
    from pyft8_amd import synth
    cases = [(("CQ", "G4ABC", "IO91"), 1234.5), (("EA5OL", "IK4LZH", "R-07"), 2711.25)]
    n = 79 * 1920
    idx = np.unique(np.concatenate([np.arange(0, 2400), np.arange(n - 2400, n), np.arange(0, n, 16)]))
    out = {"idx": idx.astype(np.int32)}
    for k, (msg, f0) in enumerate(cases):
        symbols = tx.encode_bits77(synth.pack77(*msg)) if hasattr(tx, "encode_bits77") else synth.tones79(synth.pack77(*msg))
        symbols = [int(x) for x in symbols]
        assert symbols == synth.tones79(synth.pack77(*msg))
        wf = tx.symbols_to_complex_audio(symbols, f_base=f0)
        assert len(wf) == n
        out[f"tones{k}"] = np.array(symbols, np.uint8)
        out[f"f0_{k}"] = np.float64(f0)
        out[f"wf{k}"] = wf[idx].astype(np.complex128)
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "tx_waveform.npz"), **out)
    print("tests/golden/tx_waveform.npz:", len(idx), "samples per case, numpy", np.__version__)


if __name__ == "__main__":
    main()
