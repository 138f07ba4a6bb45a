#!/bin/bash
# round 6, GPU session 33: kernel stats of one exact pass of 28 images (one pass quantum) on the final tree
bash tools/prof_exact.sh 28 2>&1 | grep -v amdgpu.ids | tail -24 | cut -c1-200
