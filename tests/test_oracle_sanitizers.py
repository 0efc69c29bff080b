"""the CPU spec-oracle under AddressSanitizer + UndefinedBehaviorSanitizer.

``oracle/Makefile`` target ``asan`` builds ``libsad_oracle_asan.so`` (-fsanitize=address,undefined,
same arithmetic flags, so the same bits); a child process with libasan preloaded loads it through
``SAD_ORACLE_LIB`` and runs the whole oracle suite — golden vectors, hand-checkable edge cases and
the numpy / cKDTree / torch cross-checks — plus the padding-skip variant.  Any out-of-bounds access,
misaligned load or signed overflow in the oracle aborts the child.  CPU only: sanitizers never run
on the GPU box's device code (not available on this pool)."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r"""
import os, sys
sys.path.insert(0, os.environ["SAD_ROOT"])
import numpy as np
import oracle
assert oracle._SO.endswith("libsad_oracle_asan.so"), oracle._SO
oracle.lib()
assert "libsad_oracle_asan.so" in open("/proc/self/maps").read(), "sanitizer build not mapped"
# padding-skip variant == dense variant, under the sanitizers
from sad_amd import config, synth
cfg = config.TINY
w = synth.make_weights(cfg, 0)
pts = synth.make_tiny_batch(3, 2, cfg.n_points)
a = oracle.detector_forward(pts, cfg, w)
b = oracle.detector_forward(pts, cfg, w, skip_padding=True)
assert np.array_equal(a, b)
import pytest
rc = pytest.main(["-x", "-q", "-p", "no:cacheprovider", os.path.join(os.environ["SAD_ROOT"], "tests", "test_oracle.py")])
print("SANITIZED_ORACLE_RC", int(rc))
sys.exit(int(rc))
"""


def _runtime(name):
    out = subprocess.run(["gcc", f"-print-file-name={name}"], capture_output=True, text=True).stdout.strip()
    return out if os.path.isabs(out) and os.path.exists(out) else None


def test_oracle_suite_under_asan_ubsan():
    asan = _runtime("libasan.so")
    assert asan, "gcc's libasan.so not found"
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "asan"], stdout=subprocess.DEVNULL)
    so = os.path.join(ROOT, "oracle", "libsad_oracle_asan.so")
    env = dict(os.environ, LD_PRELOAD=asan, SAD_ORACLE_LIB=so, SAD_ROOT=ROOT,
               ASAN_OPTIONS="detect_leaks=0:abort_on_error=1", UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1",
               OMP_NUM_THREADS="4")
    p = subprocess.run([sys.executable, "-c", CHILD], env=env, capture_output=True, text=True, timeout=900)
    tail = (p.stdout + p.stderr)[-4000:]
    assert p.returncode == 0, tail
    assert "SANITIZED_ORACLE_RC 0" in p.stdout, tail
    assert "ERROR: AddressSanitizer" not in tail and "runtime error:" not in tail, tail
