"""bench.py's own launcher and shard arithmetic (no GPU): `python bench.py --gpus N` starts N fresh ranks itself."""
import json
import os
import sys

import pytest

from conftest import REPO

sys.path.insert(0, REPO)
import bench  # noqa: E402


def test_shard_sizes():
    assert bench.shard_sizes(65536, 8, "strong") == (8192, 65536)   # the north-star shape: 65 536 total
    assert bench.shard_sizes(65536, 1, "strong") == (65536, 65536)
    assert bench.shard_sizes(65536, 8, "weak") == (65536, 524288)
    with pytest.raises(ValueError):
        bench.shard_sizes(65536, 3, "strong")


def test_defaults_are_the_baseline_metric():
    a = bench.parse([])
    assert (a.gpus, a.envs, a.scaling, a.dynamics, a.graph, a.ppo) == (1, 65536, "strong", 1, 0, 0)
    assert bench.GRAD_BUCKET_FLOATS * 4 == 42555508  # the 42.56 MB gradient bucket of SURVEY 8(e)


@pytest.mark.parametrize("n", [2, 3])
def test_launcher_starts_ranks_with_a_working_rendezvous(n):
    rc, out = bench.launch_ranks(n, ["--x", "1"], script=os.path.join(REPO, "tests", "_rank_probe.py"))
    assert rc == 0
    r = json.loads(out.strip().splitlines()[-1])
    assert r["world"] == n and r["sum"] == n * (n + 1) / 2 and r["argv"] == ["--x", "1"]
    assert r["local_rank"] == "0" and r["addr"] == "127.0.0.1"


def test_launcher_reports_a_failing_rank(tmp_path):
    bad = tmp_path / "bad.py"
    bad.write_text("import os, sys\nsys.exit(3 if os.environ['RANK'] == '1' else 0)\n")
    rc, _ = bench.launch_ranks(2, [], script=str(bad))
    assert rc == 3


def test_launcher_stops_the_siblings_of_a_rank_that_dies_after_the_rendezvous():
    """A rank that dies AFTER init_process_group leaves the others inside a collective; the launcher must return that rank's
    code within seconds instead of waiting for the gloo / NCCL timeout."""
    import time
    t0 = time.time()
    rc, out = bench.launch_ranks(3, [], script=os.path.join(REPO, "tests", "_rank_die_probe.py"))
    assert rc == 3 and "unreachable" not in out
    assert time.time() - t0 < 60.0


def test_launcher_deadline(tmp_path):
    slow = tmp_path / "slow.py"
    slow.write_text("import time\ntime.sleep(600)\n")
    import time
    t0 = time.time()
    rc, _ = bench.launch_ranks(2, [], script=str(slow), deadline_s=1.0)
    assert rc == 124 and time.time() - t0 < 30.0
