#!/bin/bash
# round 6, GPU session 22: rocprofv3 kernel stats + FETCH / WRITE passes of the bench command on the final tree (fc1's tail is a gemm_mid launch now)
bash tools/prof_bench.sh r06b 2>&1 | grep -v amdgpu.ids | tail -70
