"""How the CPU oracle scales over the GPU box's host threads (baseline only): rays/s of killeroo-simple 1920x1080, one pixel sample
per pixel, at 1 .. os.cpu_count() threads, next to what the box says about its CPUs (cgroup quota, SMT).
usage: python tools/cpu_scaling.py [out.json]"""
import json
import os
import subprocess
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "tests"))
import __graft_entry__ as ge  # noqa: E402

b = ge._load_binding()
import oracle_binding as ob  # noqa: E402


def read(path):
    try:
        return open(path).read().strip()
    except OSError:
        return None


def sh(cmd):
    try:
        return subprocess.run(cmd, shell=True, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True, timeout=20).stdout.strip()
    except Exception:
        return None


info = {"os_cpu_count": os.cpu_count(), "sched_getaffinity": len(os.sched_getaffinity(0)), "cgroup_cpu_max": read("/sys/fs/cgroup/cpu.max"),
        "cgroup_v1_quota": read("/sys/fs/cgroup/cpu/cpu.cfs_quota_us"), "cgroup_v1_period": read("/sys/fs/cgroup/cpu/cpu.cfs_period_us"),
        "lscpu": sh("lscpu | grep -E 'Model name|Socket|Core|Thread|^CPU\\(s\\)|NUMA node\\(s\\)|L3'"), "loadavg": read("/proc/loadavg")}
scene = b.HostScene(xres=1920, yres=1080, spp=64)
orc = ob.Oracle()
n = os.cpu_count() or 1
rows = []
threads = sorted({t for t in (1, 2, 4, 8, 16, 32, 64, 128, 192, 256, n) if t <= n})
for t in threads[::-1]:
    # one pixel sample per pixel (k = 1): 2.07 M camera samples, ~12 M rays; a single thread gets a quarter of the tiles' rows less: same work
    _, st = orc.render(scene, trig_mode=ob.TRIG_LIBM, threads=t, k_begin=1, k_end=2)
    rays = st["regular_rays"] + st["shadow_rays"]
    rows.append({"threads": t, "seconds": round(st["seconds"], 3), "mray_per_s": round(rays / st["seconds"] / 1e6, 3)})
    print(rows[-1], flush=True)
    if st["seconds"] > 40:
        break
rows.sort(key=lambda r: r["threads"])
base = rows[0]
for r in rows:
    r["speedup_vs_fewest"] = round(r["mray_per_s"] / base["mray_per_s"] * base["threads"], 2)
out = {"workload": "killeroo-simple 1920x1080, pixel sample k = 1 of 64, libm trig, tile self-scheduling (16x16 tiles, 8160 of them)", "host": info, "curve": rows}
print(json.dumps(out))
if len(sys.argv) > 1:
    json.dump(out, open(sys.argv[1], "w"), indent=1)
