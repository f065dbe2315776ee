import json
import os
import re
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def lbm():
    import mpilattice_boltzmann_amd as pkg
    pkg.build()
    return pkg


@pytest.fixture(scope="session")
def oracle():
    import oracle_lib
    oracle_lib.lib()
    return oracle_lib


@pytest.fixture(scope="session")
def digests():
    with open(os.path.join(GOLDEN, "digests.json")) as fh:
        return json.load(fh)


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


def deck_paths(name, digests):
    d = digests[name]
    return os.path.join(GOLDEN, "decks", d["params"]), os.path.join(GOLDEN, "decks", d["obstacles"])


# ---- rank processes started by a test (torch.distributed.run) ---------------------------------------------------------------
# What may be tried a second time, and what may not.  A set of rank processes can fail to come up for reasons that have
# nothing to do with the code under test: the c10d store or gloo's full-mesh connect not getting a socket in time on a
# loaded box.  That — and only that — is run once more on a new port.  Anything that happens after a rank has joined the
# group (the worker prints "RANK <r> UP" then) is a verdict: an LbmError (which includes the peer-to-peer loop's bounded
# waits, "did not arrive in time": the visible form of a hang), an assert, a GPU fault, a signal.  Never retried.
RENDEZVOUS_TEXT = ("RendezvousConnectionError", "RendezvousTimeoutError", "DistNetworkError", "DistStoreError", "connectFullMesh",
                   "Connection refused", "Connection reset by peer", "Address already in use", "EADDRINUSE",
                   "The client socket has timed out", "failed to connect to", "Gloo connectFullMesh failed")
VERDICT_TEXT = ("LbmError", "lbm_p2p", "lbm_comm", "did not arrive", "out of step", "FAILED", "AssertionError", "Memory access fault",
                "HSA_STATUS", "hipError", "Segmentation fault", "core dumped")
ARTIFACTS = os.path.join(ROOT, "gpurun_out", "test_artifacts")     # gpurun merges gpurun_out/ back: the logs survive the box


def is_rendezvous_failure(returncode, stdout, stderr):
    """True only for a set of rank processes that never formed its group: non-zero exit, no rank got as far as
    "RANK <r> UP", rendezvous / socket text on stderr, and nothing that reads like a verdict of the code under test."""
    if returncode == 0:
        return False
    text = stdout + "\n" + stderr
    if " UP" in stdout and any(line.startswith("RANK ") and line.rstrip().endswith(" UP") for line in stdout.splitlines()):
        return False
    if any(v in text for v in VERDICT_TEXT):
        return False
    # torch.distributed.run's summary: the FIRST rank to fail ("Root Cause") must have exited by itself; a signal there is a
    # crash (the ranks the agent then stops with SIGTERM are listed under "Other Failures" and do not count)
    root = text.split("Root Cause", 1)[1] if "Root Cause" in text else ""
    m = re.search(r"exitcode\s*:\s*(-?\d+)", root)
    if m and int(m.group(1)) < 0:
        return False
    return any(r in text for r in RENDEZVOUS_TEXT)


def run_rank_processes(make_cmd, env, tag, timeout=1200):
    """subprocess.run of make_cmd(port) with both attempts' complete output kept under gpurun_out/test_artifacts/<tag>.attemptN.log
    (on a pass as well).  At most one more attempt, and only after is_rendezvous_failure()."""
    import socket
    import subprocess
    os.makedirs(ARTIFACTS, exist_ok=True)
    r = None
    for attempt in (1, 2):
        with socket.socket() as sock:
            sock.bind(("127.0.0.1", 0))
            port = sock.getsockname()[1]
        cmd = make_cmd(port)
        r = subprocess.run(cmd, capture_output=True, text=True, timeout=timeout, env=env)
        with open(os.path.join(ARTIFACTS, f"{tag}.attempt{attempt}.log"), "w") as fh:
            fh.write(f"$ {' '.join(cmd)}\nreturncode {r.returncode}\n---- stdout ----\n{r.stdout}\n---- stderr ----\n{r.stderr}\n")
        if attempt == 1 and is_rendezvous_failure(r.returncode, r.stdout, r.stderr):
            print(f"{tag}: the rank processes never formed their group (attempt 1 kept in {ARTIFACTS}); once more on a new port")
            continue
        break
    return r
