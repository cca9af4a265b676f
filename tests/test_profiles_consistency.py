"""The committed evidence is reproducible from the committed raw summaries: profiles/pmc_traffic.json (which bench.py reads for
`roofline.traffic`) == tools/pmc_traffic.py over profiles/r06/*_pmc_{FETCH,WRITE}_SIZE.txt."""
import json
import os
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _strip(o):
    if isinstance(o, dict):
        return {k: _strip(v) for k, v in o.items() if k not in ("source", "traffic_source")}
    return o


def test_pmc_traffic_json_is_what_the_tool_derives_from_the_committed_passes(tmp_path):
    src = os.path.join(ROOT, "profiles", "r06")
    dst = tmp_path / "profiles" / "r06"
    dst.mkdir(parents=True)
    for f in os.listdir(src):
        if "_pmc_FETCH_SIZE" in f or "_pmc_WRITE_SIZE" in f:
            shutil.copy(os.path.join(src, f), dst / f)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "pmc_traffic.py"), str(dst)], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr[-2000:]
    new = json.load(open(tmp_path / "profiles" / "pmc_traffic.json"))
    old = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))
    assert _strip(new) == _strip(old)
    wf = old["ffhq_B16_256_f16x3"]["whole_forward"]
    assert 30 < wf["GB_per_forward"] < 60 and abs(2 * wf["fetch_GB_raw"] + wf["write_GB"] - wf["GB_per_forward"]) < 0.05


def test_bench_reads_the_committed_traffic_figures():
    """bench.py's `roofline.traffic` is the 3x3 class entry of profiles/pmc_traffic.json (it cannot run rocprofv3 on itself)."""
    sys.path.insert(0, ROOT)
    import bench
    old = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))
    src = open(os.path.join(ROOT, "bench.py")).read()
    assert "pmc_traffic.json" in src
    assert old["ffhq_B16_256_f16x3"]["bytes_per_launch"] > 1e8
    assert hasattr(bench, "main") or hasattr(bench, "conv_roofline")
