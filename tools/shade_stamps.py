"""Where a k_shade wavefront's cycles go: the diagnostic build (-DIILE_SHADE_STAMPS: s_memtime stamps between the sections of a
round, tools/build_variant.sh stamps "kernels_shade api" "-DIILE_SHADE_STAMPS") renders the bench frame once and prints the
per-section share of the waves' cycles. The stamps themselves cost ~10 %: shares, not absolute times.
usage: IILE_GPU_LIB=pbrt-v3-iile_amd/lib/variants/libiile_gpu_stamps.so python tools/shade_stamps.py"""
import json
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import __graft_entry__ as ge  # noqa: E402

b = ge._load_binding()
scene = b.HostScene(xres=1920, yres=1080, spp=64)
gpu = b.GpuScene(scene)
gpu.render()
_, st = gpu.render(want_stats=True, time_kernels=2)
names = ["regroup chunk", "loads + 4 Halton dims", "interaction + BSDF set-up", "light-sampling half", "BSDF-sampling half", "NEE record stores",
         "continuation (2 Halton dims, Sample_f, RR)", "next-ray store"]
cyc = [int(x) for x in st["path_length"]]
tot = sum(cyc) or 1
print(json.dumps({"ms_shade_one_stream": st["ms_shade"], "wave_cycles_total": tot,
                  "sections": {n: round(c / tot, 4) for n, c in zip(names, cyc)}}, indent=1))
