"""Host sanitizer targets: the CPU oracle and the product's pure-host C++ (native message layer, tone
encoder) built with gcc/g++ -fsanitize=address,undefined and run on real inputs.  GPU code cannot run under sanitizers on this
pool; the kernels are covered by the bit-exact parity tests instead."""
import os
import shutil
import subprocess

import numpy as np
import pytest

from conftest import GOLDEN, ROOT, load_golden
from helpers import oracle_frame, records_from_oracle

pytestmark = pytest.mark.skipif(shutil.which("gcc") is None or shutil.which("g++") is None, reason="needs gcc/g++")


@pytest.fixture(scope="module")
def asan_bins():
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "asan"], stderr=subprocess.DEVNULL)
    b = os.path.join(ROOT, "oracle", "_build")
    return os.path.join(b, "asan_oracle"), os.path.join(b, "asan_host")


def _run(cmd):
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1")
    p = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stderr[-3000:]
    assert "AddressSanitizer" not in p.stderr and "runtime error" not in p.stderr and "LeakSanitizer" not in p.stderr, p.stderr[-3000:]
    return p.stdout


def test_oracle_under_asan_ubsan(asan_bins):
    out = _run([asan_bins[0], os.path.join(GOLDEN, "test_08.wav"), os.path.join(GOLDEN, "test_09.wav")])
    assert "silence: 0 messages" in out
    assert "test_08.wav: 180000 samples, 21 messages (reference knobs)" in out and "test_09.wav: 180000 samples, 20 messages" in out


def test_host_message_layer_under_asan_ubsan(asan_bins, tmp_path):
    from pyft8_amd import _lib
    rec, n, ev, nev = records_from_oracle(oracle_frame(load_golden("test_09")[0]))
    dump = tmp_path / "frame.bin"
    with open(dump, "wb") as f:
        np.array([n, min(nev, _lib.EVENT_CAP)], np.int32).tofile(f)
        rec[:n].tofile(f)
        ev[:min(nev, _lib.EVENT_CAP)].tofile(f)
    out = _run([asan_bins[1], str(dump)])
    assert f"frame dump: {n} candidates" in out and "20 messages per frame" in out and "random words:" in out
