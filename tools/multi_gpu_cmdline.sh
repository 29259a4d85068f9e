#!/bin/bash
# Prints the exact command lines of an N-GPU render / bench on one node (default N = 8); run them from the repo root.
#   tools/multi_gpu_cmdline.sh [N] [scene.pbrt]
N=${1:-8}
SCENE=${2:-scenes/killeroo-simple.pbrt}
JOB=1$RANDOM$RANDOM   # (never starts with 0; iile_pbrt reads it as a decimal number > 0)
echo "# C++ host, one process per GPU (rank R uses GPU R); rank 0 writes the image:"
echo "export HSA_ENABLE_IPC_MODE_LEGACY=0"
for ((r = 0; r < N; r++)); do
  echo "pbrt-v3-iile_amd/lib/iile_pbrt $SCENE --xres 1920 --yres 1080 --spp 1024 --outfile killeroo_1024spp.exr --stats --gpurank $r/$N --rendezvous /tmp/iile_rv_$JOB --job $JOB &"
done
echo "wait"
echo
echo "# bench (the driver's contract): the fixed 1080p x 1024 spp frame of BASELINE config 3 split over the ranks (\"scaling\": \"strong\")"
echo "python -m torch.distributed.run --nnodes=1 --nproc-per-node $N --master-addr 127.0.0.1 --master-port 29500 bench.py --gpus $N --steps 3 --warmup 1"
echo
echo "# the IISPT frame (BASELINE config 5) over the same ranks: tasks by their number, direct passes in blocks, the film monitors summed on rank 0"
echo "python -m torch.distributed.run --nnodes=1 --nproc-per-node $N --master-addr 127.0.0.1 --master-port 29501 bench.py --workload iispt --gpus $N --steps 3 --warmup 1"
JOB2=2$RANDOM$RANDOM
for ((r = 0; r < N; r++)); do
  echo "pbrt-v3-iile_amd/lib/iile_pbrt $SCENE --xres 1920 --yres 1080 --integrator iispt --iisptNet=weights.iilenet --iileIndirect=220 --outfile killeroo_iispt.exr --gpurank $r/$N --rendezvous /tmp/iile_rv_$JOB2 --job $JOB2 &"
done
echo "wait"
