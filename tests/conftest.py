import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run by the driver with -m gpu)")


@pytest.fixture(scope="session")
def root():
    return ROOT


@pytest.fixture(scope="session")
def trained_blob(root):
    return os.path.join(root, "weights", "tandem_va.tdmw")


@pytest.fixture
def parity_hooks():
    """Run the test against the PARITY build (tandem_amd/libdr_mi355x_hooks.so = the same sources with -DDR_PARITY_HOOKS): the product
    library does not contain superseded kernel generations, so the cases that compare generations (DR_RAYCAST_V1, DR_RAYCAST_SAMPLER,
    DR_OUT3_FOLDED=0, DR_NO_SKIP_FUSION, ...) load this one for their duration."""
    from tandem_amd import _lib
    if not os.path.isfile(_lib.HOOKS_LIB_PATH):
        pytest.skip("tandem_amd/libdr_mi355x_hooks.so not built")
    _lib.switch(_lib.HOOKS_LIB_PATH)
    try:
        yield _lib.HOOKS_LIB_PATH
    finally:
        _lib.switch(None)


def launch_gloo_ranks(script_path, world=2, timeout=600, attempts=3, extra_env=None):
    """Run `script_path` as `world` processes with the torch.distributed.run environment on 127.0.0.1 and return each
    rank's last stdout line parsed as JSON, sorted by rank.  The rendezvous port is found by bind-and-release, which
    another process can win in between: a failed rendezvous is retried on a fresh port."""
    import json
    import os
    import socket
    import subprocess
    import sys
    last = ""
    for _ in range(attempts):
        s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
        procs = []
        for r in range(world):
            env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
            env.update(extra_env or {})
            procs.append(subprocess.Popen([sys.executable, str(script_path)], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
        outs, ok = [], True
        for p in procs:
            try:
                o, e = p.communicate(timeout=timeout)
            except subprocess.TimeoutExpired:
                p.kill(); o, e = p.communicate()
                ok = False
            if p.returncode != 0:
                ok, last = False, e[-3000:]
            else:
                outs.append(json.loads(o.strip().splitlines()[-1]))
        if ok:
            return sorted(outs, key=lambda d: d["rank"])
    raise AssertionError("two-rank run failed %d times; last stderr:\n%s" % (attempts, last))
