"""Host-side logic of bench.py that needs no GPU."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_traffic_is_reported_only_for_the_profiled_instantiation():
    """roofline.traffic comes from committed counter passes: bench.pmc_traffic must hand it out only for the workload
    and the exact kernel instantiation those passes were taken on."""
    import json

    sys.path.insert(0, ROOT)
    import bench

    prof = json.load(open(os.path.join(ROOT, bench.TRAFFIC_PROFILE)))
    rows, cols = prof["workload"]["rows"], prof["workload"]["cols"]
    timed = prof["timed_kernel"]
    got, src = bench.pmc_traffic(rows, cols, rows, timed)
    assert src == bench.TRAFFIC_PROFILE and 0.95 * rows * cols * 8 < got < 1.1 * rows * cols * 8
    other = timed.replace("false, false>", "true, false>")  # theta in the kernarg: a different instantiation
    assert other != timed and bench.pmc_traffic(rows, cols, rows, other) == (None, None)
    assert bench.pmc_traffic(rows // 2, cols, rows // 2, timed) == (None, None)
    assert bench.pmc_traffic(rows, cols, rows // 2, timed) == (None, None)  # a shard of the job
