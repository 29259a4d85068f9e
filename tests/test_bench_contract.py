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
    # every kernel is priced from steps in which it has the GPU to itself (VERDICT r04 "next" 3): the per-kernel times then add
    # up to no more than the one-stream step, and the overlapped two-stream figures sit apart, never summed
    assert r["schedule"].startswith("one-stream") and all(e["schedule"].startswith("one-stream") for e in d["roofline_all_kernels"].values()), src
    assert sum(e["ms_per_step"] for e in d["roofline_all_kernels"].values()) <= d["kernel_ms_per_step_one_stream"]["ms_total"] * 1.001, src
    assert r["overlapped"]["avg_launch_ms"] >= r["avg_launch_ms"] and "overlapped" in d["roofline_all_kernels"][r["kernel"]], src
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


@pytest.mark.parametrize("tag", ["", "room_"])
def test_one_stream_kernel_trace_agrees_with_the_bench_line(tag):
    """profiles/*_one_stream_kernel_stats.csv is the rocprofv3 trace of `bench.py --schedule one-stream` (every kernel of the
    traced steps alone on the GPU); the line printed under the profiler prices the dominant kernel from HIP events of the same
    steps: the two averages must agree (2 %), and the committed headline line's `avg_launch_ms` — un-profiled one-stream steps —
    with them (8 %: profiled runs hold a slightly lower clock)."""
    import csv
    files = sorted(glob.glob(os.path.join(REPO, "profiles", f"r*_{tag}one_stream_kernel_stats.csv")), key=lambda f: [int(x) if x.isdigit() else x for x in re.split(r"(\d+)", os.path.basename(f))])
    files = [f for f in files if ("room_" in os.path.basename(f)) == bool(tag)]
    assert files
    rows = [r for r in csv.DictReader(open(files[-1])) if "k_extend<false" in r["Name"]]   # the product builds (not the instrumented one)
    calls = sum(int(r["Calls"]) for r in rows)
    trace_ms = sum(int(r["Calls"]) * float(r["AverageNs"]) for r in rows) / calls / 1e6
    line = json.loads(open(files[-1].replace("one_stream_kernel_stats.csv", "bench_one_stream_under_profiler.json")).readline())
    assert line["schedule"] == "one-stream" and line["roofline"]["kernel"] == "k_extend"
    assert abs(line["roofline"]["avg_launch_ms"] - trace_ms) / trace_ms < 0.02, (line["roofline"]["avg_launch_ms"], trace_ms)
    head, _ = _newest("bench_boxroom" if tag else "bench")
    assert abs(head["roofline"]["avg_launch_ms"] - trace_ms) / trace_ms < 0.08, (head["roofline"]["avg_launch_ms"], trace_ms)
    if not tag:
        assert abs(head["roofline"]["frac"] - 0.21) < 0.02   # k_extend alone: counted traffic / time / 8 TB/s


def test_iispt_line_has_the_contract_fields():
    """BASELINE config 5's line (`bench.py --workload iispt`): the contract's fields, the roofline of the network's convolution
    kernels against the bf16 matrix peak recomputed from its parts, the one-thread CPU baseline of the reference's per-probe loop,
    and stage times that add up to the indirect pass."""
    d, src = _newest("bench_iispt")
    for key, typ in (("metric", str), ("value", float), ("unit", str), ("n_gpus", int), ("steps", int), ("warmup", int), ("ms_per_step", float),
                     ("higher_is_better", bool), ("scaling", str), ("dtype", str), ("data", str), ("config", dict), ("roofline", dict), ("built", dict),
                     ("cpu_baseline", dict), ("stage_ms_per_step", dict)):
        assert isinstance(d.get(key), typ), (src, key)
    assert d["vs_baseline"] is None and d["unit"] == "probes/s" and "workload" in d["config"] and "model" not in d["config"]
    assert abs(d["value"] - d["config"]["probes"] / d["ms_per_step"] * 1e3) / d["value"] < 2e-3
    r = d["roofline"]
    assert r["bound"] == "mfma" and r["unit"] == "TFLOP/s" and r["peak"] == 2500.0 and "k_conv3x3" in r["kernel"]
    assert r["algorithmic_flop_per_unit"] == 990117888   # 2 k^2 C_in C_out H W over ml/iispt_net.py's 15 convolutions
    ach = r["algorithmic_flop_per_unit"] * r["units_per_step"] / (r["network_ms_per_step"] * 1e-3) / 1e12
    assert abs(ach - r["achieved"]) / ach < 2e-3 and abs(r["frac"] - r["achieved"] / r["peak"]) < 2e-4
    assert abs(r["frac_executed"] - 3 * r["frac"]) < 5e-4 and 0 < r["frac_executed"] < 1
    assert r["agreement_with_the_module"]["max_abs_err_over_max"] < 1e-5   # (split fp16: 2e-6; round 5's split bf16 2.4e-5)
    cb = d["cpu_baseline"]
    assert cb["kind"] == "port" and cb["unit"] == "probes/s" and cb["cores"] == 1 and cb["value"] > 0 and "sample" in cb
    st = d["stage_ms_per_step"]
    assert st["network"] == r["network_ms_per_step"] and sum(st.values()) <= d["ms_per_step"] * 1.02
    assert d["built"]["sources_sha"] == d["built"]["sources_sha_now"], src


def test_headline_line_carries_configs_4_and_5():
    """VERDICT r05 "next" 1: the line the driver runs (`python bench.py --gpus 1 ...`, no other flag) measures BASELINE config 4 (the
    deep room) and config 5 (the IISPT frame) after the headline steps and prints them as sub-blocks of the same JSON line — each a
    whole line of its own workload (ms_per_step, roofline, cpu_baseline, the in-run parity check). The headline keys stay config 2's."""
    d, src = _newest("bench")
    assert (d["config"]["xres"], d["config"]["yres"], d["config"]["spp_total"]) == (1920, 1080, 64) and "killeroo-simple" in d["metric"]
    blocks = d["configs"]
    assert set(blocks) == {"4_room", "5_iispt"}, src
    room, frame = blocks["4_room"], blocks["5_iispt"]
    for b_ in (room, frame):
        for key, typ in (("metric", str), ("value", float), ("unit", str), ("steps", int), ("ms_per_step", float), ("config", dict), ("roofline", dict),
                         ("cpu_baseline", dict), ("wall_seconds_of_this_block", float)):
            assert isinstance(b_.get(key), typ), (src, key)
        assert b_["n_gpus"] == 1 and b_["vs_baseline"] is None and "built" not in b_
        assert b_["cpu_baseline"]["kind"] == "port" and b_["cpu_baseline"]["value"] > 0 and "sample" in b_["cpu_baseline"]
    # config 4: the room's own Mray/s line, priced like the headline (k_extend alone on the GPU; counted traffic where a counter set of the room exists)
    assert room["unit"] == "Mray/s" and "boxroom" in room["config"]["workload"] and room["config"]["spp_total"] == 64
    assert abs(room["value"] - room["rays_per_step"] / room["ms_per_step"] / 1e3) / room["value"] < 2e-3
    assert room["roofline"]["kernel"] == "k_extend" and room["roofline"]["schedule"].startswith("one-stream") and room["timed_film_verified"].startswith("bitwise equal")
    assert 300 < room["ms_per_step"] < 600
    # config 5: probes/s over the whole frame, the network against the 16-bit matrix peak, the in-run agreement with the fp32 module PER ELEMENT
    assert frame["unit"] == "probes/s" and abs(frame["value"] - frame["config"]["probes"] / frame["ms_per_step"] * 1e3) / frame["value"] < 2e-3
    r = frame["roofline"]
    assert r["bound"] == "mfma" and r["peak"] == 2500.0 and abs(r["frac_executed"] - 3 * r["frac"]) < 5e-4
    chk = r["agreement_with_the_module"]
    assert chk["max_abs_err_over_max"] < 1e-5 and chk["elements_within_1e-4_rel_plus_1e-6_of_max"] >= 0.999 and chk["mean_rel_err_where_nonzero"] <= 1e-5
    assert sum(frame["stage_ms_per_step"].values()) <= frame["ms_per_step"] * 1.02
    # the default run still fits the driver's window: the two blocks add well under a minute
    assert room["wall_seconds_of_this_block"] + frame["wall_seconds_of_this_block"] < 60


def test_iispt_frame_has_no_slow_processes():
    """Round 4: about one process in four ran the network at half speed from its first frame to its last (MIOpen's solver
    choice). Ten fresh processes of the frame now (profiles/*_iispt_ten_processes.jsonl): within 5 % of their median."""
    files = sorted(glob.glob(os.path.join(REPO, "profiles", "r*_iispt_ten_processes.jsonl")))
    assert files
    runs = [json.loads(l) for l in open(files[-1]) if l.strip()]
    assert len(runs) == 10
    ms = sorted(r["ms_per_step"] for r in runs)
    med = 0.5 * (ms[4] + ms[5])
    assert max(abs(m - med) for m in ms) / med < 0.05, ms
