"""The bench line's contract (the driver parses it; the judge recomputes from it), checked on the committed evidence lines of the
newest set under profiles/ — no GPU needed: the required keys and types, the metric of BASELINE.json on its configuration, and
the arithmetic a reader would redo (value = rays / time; roofline.achieved, frac_algorithmic and frac from their parts)."""
import glob
import json
import os
import re

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _newest(name):
    files = sorted(glob.glob(os.path.join(REPO, "profiles", f"r*_{name}.json")), key=lambda f: [int(x) if x.isdigit() else x for x in re.split(r"(\d+)", os.path.basename(f))])
    assert files, name
    return json.loads(open(files[-1]).readline()), os.path.basename(files[-1])


@pytest.mark.parametrize("name", ["bench", "bench_boxroom", "bench_torchrun1"])
def test_bench_line_has_the_contract_fields(name):
    d, src = _newest(name)
    for key, typ in (("metric", str), ("value", float), ("unit", str), ("n_gpus", int), ("steps", int), ("warmup", int), ("ms_per_step", float),
                     ("higher_is_better", bool), ("scaling", str), ("dtype", str), ("data", str), ("config", dict), ("roofline", dict), ("built", dict),
                     ("per_rank", dict)):
        assert isinstance(d.get(key), typ), (src, key)
    assert "vs_baseline" in d and d["vs_baseline"] is None  # BASELINE.md holds no published number for this metric
    assert d["unit"] == "Mray/s" and d["higher_is_better"] is True and d["dtype"] == "f32" and d["scaling"] in ("weak", "strong")
    assert "workload" in d["config"] and "model" not in d["config"]
    # value is whole-job throughput: rays traced per step / step time
    assert abs(d["value"] - d["rays_per_step"] / d["ms_per_step"] / 1e3) / d["value"] < 2e-3, src
    r = d["roofline"]
    for key in ("bound", "achieved", "peak", "unit", "frac", "traffic", "kernel"):
        assert key in r, (src, key)
    assert r["bound"] in ("hbm", "vmem", "registers+vmem", "mfma") and r["unit"] in ("GB/s", "TFLOP/s")
    # `achieved` is SURVEY.md 8(d)'s algorithmic rate (bytes of the reference's layout per launch / launch time) and
    # `frac_algorithmic` = achieved / peak (it may exceed 1: the scene is cache resident); `frac` is the judge's one definition for
    # every kernel, counted traffic / launch time / peak (VERDICT r03 "next" 3), present where counters of this step size exist
    assert abs(r["frac_algorithmic"] - r["achieved"] / r["peak"]) < 2e-3, src
    assert abs(r["achieved"] - r["algorithmic_bytes_per_launch"] / (r["avg_launch_ms"] * 1e-3) / 1e9) / r["achieved"] < 2e-3, src
    if r["frac"] is not None:
        assert abs(r["frac"] - r["traffic"] / (r["avg_launch_ms"] * 1e-3) / 1e9 / r["peak"]) < 2e-3, src
        assert 0 < r["frac"] < 1
    else:
        assert r["traffic"] is None
    b = d["built"]
    assert b["sources_sha"] == b["sources_sha_now"], (src, "the evidence line was not made from the sources it was built from")


def test_headline_line_is_baselines_configuration():
    d, src = _newest("bench")
    base = json.load(open(os.path.join(REPO, "BASELINE.json")))
    assert "killeroo-simple" in d["metric"] and "Mray/s" in base.get("metric", "Mray/s")
    c = d["config"]
    assert (c["xres"], c["yres"], c["spp_total"]) == (1920, 1080, 64) and d["n_gpus"] == 1
    cb = d["cpu_baseline"]
    for key in ("value", "unit", "cores", "kind", "sample"):
        assert key in cb, key
    assert cb["kind"] in ("port", "reference") and cb["unit"] == "Mray/s" and cb["cores"] >= 1 and cb["value"] > 0
    assert d["timed_film_verified"].startswith("bitwise equal")
