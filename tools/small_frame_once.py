import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import __graft_entry__ as ge
b = ge._load_binding()
scene = b.HostScene(xres=128, yres=96, spp=4)
gpu = b.GpuScene(scene)
for _ in range(3):
    gpu.render()
