#!/bin/bash
# What the GPU box's host really offers: cgroup CPU quota, visible cpus, and how a fixed fp32 GEMM load scales over processes.
echo "nproc $(nproc)  cpu.max $(cat /sys/fs/cgroup/cpu.max 2>/dev/null)  cpuset $(cat /sys/fs/cgroup/cpuset.cpus.effective 2>/dev/null)"
cat /sys/fs/cgroup/cpu.stat 2>/dev/null | head -8
python - <<'PY'
import os, time, subprocess, sys
code = "import torch,time,os;torch.set_num_threads(int(os.environ['T']));a=torch.randn(4096,4096);b=torch.randn(4096,4096);a@b;t=time.time();[a@b for _ in range(int(os.environ['R']))];print((time.time()-t))"
for procs, thr in ((1, 16), (4, 16), (16, 16), (16, 8), (1, 64)):
    ps = []
    t0 = time.time()
    for p in range(procs):
        env = dict(os.environ, T=str(thr), R="20", OMP_NUM_THREADS=str(thr))
        ps.append(subprocess.Popen(["taskset", "-c", f"{p*thr}-{p*thr+thr-1}", sys.executable, "-c", code], env=env, stdout=subprocess.PIPE))
    outs = [float(p.communicate()[0]) for p in ps]
    fl = procs * 20 * 2 * 4096**3 / max(outs) / 1e12
    print(f"{procs} procs x {thr} threads: {fl:.2f} TFLOP/s fp32 total (slowest {max(outs):.2f} s)", flush=True)
PY
grep -i throttl /sys/fs/cgroup/cpu.stat 2>/dev/null
