"""GPU: failure path of the factorisation's in-kernel hand-offs (include/gpk.h, "info" and gpk_chain_handoff_mode).

The chain of a factorisation hands over between the library's streams through flag words that kernels write and kernels wait for;
every wait is bounded (0.5 s).  The A/B build (libgpk_exp.so, `make exp`; never loaded by the package unless GPK_LIBRARY points at
it) can WITHHOLD one of those words (GPK_FAULT_DROP_REST_FLAG=p: "rest-update of panel p done" is never written).  The call must then
still return -- within the bound, not hang --, the status word must be INT_MAX, and the host layer must say so instead of reporting a
non-positive pivot at column 2147483646."""
import os
import subprocess
import sys
import time

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
XLIB = os.path.join(ROOT, "gpflow_amd", "libgpk_exp.so")

CODE = r"""
import os, sys, time, numpy as np, torch
sys.path.insert(0, %r)
from gpflow_amd import _lib, ops
rng = np.random.default_rng(3)
n = 1024
B = rng.normal(size=(n, n)); K = B @ B.T / n + np.eye(n)
T = ops.to_device(np.vstack([K, rng.normal(size=(300, n))]))
torch.cuda.synchronize()
t0 = time.perf_counter()
_, info = ops.potrf_(T, n, zero_upper=True)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
code = int(info.cpu().numpy()[0])
msg = ""
try:
    ops.check_info(info)
except _lib.GpkError as e:
    msg = str(e)
L = T.cpu().numpy()[:n]
print("RESULT", code, round(dt, 3), float(np.abs(L @ L.T - K).max()) if code == 0 else -1.0, "|", msg)
"""


def _run(env_extra):
    env = dict(os.environ, GPK_LIBRARY=XLIB, **env_extra)
    t0 = time.perf_counter()
    r = subprocess.run([sys.executable, "-c", CODE % ROOT], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("RESULT")][0]
    head, msg = line.split("|", 1)
    _, code, dt, err = head.split()
    return int(code), float(dt), float(err), msg.strip(), time.perf_counter() - t0


@pytest.mark.skipif(not os.path.exists(XLIB), reason="the A/B library (make exp) is not built")
def test_withheld_flag_times_out_and_is_reported():
    code, dt, err, msg, _ = _run({})                                    # control: the A/B library without the fault
    assert code == 0 and err < 5e-12 and msg == ""
    code, dt, err, msg, _ = _run({"GPK_FAULT_DROP_REST_FLAG": "1"})     # "rest-update 1 done" never arrives: strip 2 polls in vain
    assert code == 2 ** 31 - 1, code
    assert 0.4 < dt < 5.0, dt                                           # one bounded wait of 0.5 s (plus whatever queued behind it)
    assert "hand-off timed out" in msg and "pivot" not in msg, msg
