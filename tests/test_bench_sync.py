"""bench.py's FileSync: the barrier / all-gather through files the encoder leg of `bench.py --gpus N` uses to keep N ranks together BEFORE any of them opens its GPU
(the nccl process group would).  World sizes 2 and 4 as separate processes on the CPU."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

RANK = r"""
import faulthandler, json, os, sys, time
faulthandler.dump_traceback_later(100, exit=True)
sys.path.insert(0, sys.argv[4])
import bench
rank, world, root = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3]
s = bench.FileSync(rank, world, root=root)
time.sleep(0.05 * (world - rank))                                   # ranks arrive in reverse order
md5 = s.exchange("0123456789abcdef" if rank == 0 else "")[0]        # the broadcast of the leg
secs = max(float(v) for v in s.exchange(repr(1.5 + rank if rank != 1 else float("inf"))))   # the max of the leg, one rank reporting a differing bitstream
again = s.exchange("r%d" % rank)
s.close()
print(json.dumps([rank, md5, repr(secs), again]))
"""


@pytest.mark.parametrize("world", [2, 4])
def test_file_sync_broadcast_max_and_cleanup(tmp_path, world):
    env = dict(os.environ, MASTER_PORT="29999", SVT_BENCH_SYNC_KEY="unit%d" % world)
    ps = [subprocess.Popen([sys.executable, "-c", RANK, str(r), str(world), str(tmp_path), ROOT], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=env)
          for r in range(world)]
    outs = []
    for p in ps:
        try:
            outs.append(p.communicate(timeout=90))
        except subprocess.TimeoutExpired:
            p.kill()
            o, e = p.communicate()
            raise AssertionError("rank hung: " + o[-500:] + e[-1500:])
    for r, (p, (out, err)) in enumerate(zip(ps, outs)):
        assert p.returncode == 0, err[-1500:]
        rank, md5, secs, again = json.loads(out.strip().splitlines()[-1])
        assert rank == r and md5 == "0123456789abcdef" and secs == "inf" and again == ["r%d" % k for k in range(world)]
    assert not os.listdir(str(tmp_path)), "rank 0 removes the directory once every rank has said goodbye"


def test_file_sync_times_out_loudly(tmp_path):
    code = ("import sys; sys.path.insert(0, sys.argv[2]); import bench\n"
            "s = bench.FileSync(0, 2, root=sys.argv[1])\n"
            "try:\n    s.exchange('x', timeout_s=0.2)\nexcept RuntimeError as e:\n    print('LOUD', e)\n")
    r = subprocess.run([sys.executable, "-c", code, str(tmp_path), ROOT], capture_output=True, text=True, timeout=120, env=dict(os.environ, SVT_BENCH_SYNC_KEY="alone"))
    assert r.returncode == 0 and "LOUD FileSync: rank 0 waited" in r.stdout, r.stdout + r.stderr[-800:]
