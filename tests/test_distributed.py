"""N>1 path on CPU: world_size 2, gloo backend.  Checks the batch sharding and the result gathers that bench.py /
dist.py use at N>1 (the GPU kernels themselves are covered by the -m gpu tests)."""
import json
import os
import socket
import subprocess
import sys

from tests.helpers import ROOT


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_sharded_decode_world2_gloo():
    port = _free_port()
    worker = os.path.join(ROOT, "tests", "dist_worker.py")
    procs = [subprocess.Popen([sys.executable, worker, str(r), "2", str(port)], stdout=subprocess.PIPE,
                              stderr=subprocess.STDOUT, text=True, cwd=ROOT) for r in range(2)]
    outs = [p.communicate(timeout=180)[0] for p in procs]
    want = [[10 * k + i, (10 * k + i) % 4 + 1] for k in range(5) for i in range(3)]
    for p, out in zip(procs, outs):
        assert p.returncode == 0, out
        line = [l for l in out.splitlines() if l.startswith("RESULT ")][-1]
        assert json.loads(line[7:]) == want
