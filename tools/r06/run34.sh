#!/bin/bash
# round 6, GPU session 34: the driver's multi-GPU launch form with one rank (torchrun env -> RCCL communicator through the C ABI)
mkdir -p gpurun_out/r06
timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --steps 4 --warmup 1 --no-extras --cpu-images 0 > gpurun_out/r06/torchrun1.json 2> gpurun_out/r06/torchrun1.err
tail -1 gpurun_out/r06/torchrun1.json | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('torchrun N=1:', round(d['value'],1), d['n_gpus'], d.get('rccl'), d['exact_pass_schedule']['same_on_every_rank'])" || tail -20 gpurun_out/r06/torchrun1.err
