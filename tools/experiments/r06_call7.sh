#!/bin/bash
# round-6 call 7: what does k_shadow's scattered read of L cost? (timing-only variant that does not read it: an upper bound on what
# "accumulate in the record's own slot, fold in k_film" could win) — killeroo-simple and the room, one-stream kernel times
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
O=$R/gpurun_out/r06_call7
mkdir -p $O
cd $R
cat > /tmp/shadow_ab.py <<'PY'
import os, sys, tempfile
sys.path.insert(0, os.environ["R"]); sys.path.insert(0, os.path.join(os.environ["R"], "tests"))
import __graft_entry__ as ge
b = ge._load_binding()
kw = {}
if sys.argv[1] == "room":
    import boxroom
    tmp = tempfile.NamedTemporaryFile("w", suffix=".pbrt", delete=False); tmp.write(boxroom.boxroom_pbrt(ico_levels=5, n_blobs=12, wall_n=64)); tmp.close()
    kw = {"path": tmp.name}
scene = b.HostScene(xres=1920, yres=1080, spp=64, **kw)
gpu = b.GpuScene(scene)
best = None
for _ in range(4):
    film, st = gpu.render(time_kernels=2)
    if best is None or st["ms_shadow"] < best["ms_shadow"]: best = st
print(sys.argv[1], os.environ.get("IILE_GPU_LIB", "default").split("_gpu_")[-1], {k: round(best[k], 2) for k in ("ms_extend", "ms_shade", "ms_shadow", "ms_mis", "ms_total")})
PY
export R
for rep in 1 2; do for w in killeroo room; do for v in default shadow_no_L_read; do
  if [ $v = default ]; then unset IILE_GPU_LIB; else export IILE_GPU_LIB=$R/pbrt-v3-iile_amd/lib/variants/libiile_gpu_$v.so; fi
  timeout 300 python3 /tmp/shadow_ab.py $w 2>&1 | tail -1
done; done; done | tee $O/shadow_ab.txt
