"""Achieved HBM bandwidth of the refinement kernels against the batch size (round 5): the bench's step refines 128 queries (one block
per (query, candidate): 640 blocks, 2.5 per CU -- a launch that is over before it reaches a steady state); what does the same kernel
reach when the launch is larger?   python tools/refine_bw.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from pigeon_amd import hip_ops, synthetic

dev = "cuda"
C, ppc, topk = 10000, 100, 5
bank = hip_ops.DeviceBank(synthetic.make_bank_device(C, ppc, seed=2, device=dev), device=dev)
g = torch.Generator(device=dev).manual_seed(3)
for B in (128, 256, 512, 1024, 2048):
    q = torch.randn((B, 4, 1024), generator=g, device=dev)
    init = torch.zeros((B, 2), dtype=torch.float64, device=dev)
    prob = torch.softmax(torch.randn((B, topk), generator=g, device=dev), dim=1).sort(dim=1, descending=True).values.contiguous()
    times, rows = [], 0
    for it in range(6):                                    # other cells every launch: nothing comes from the Infinity Cache
        cand = torch.stack([torch.randperm(C, generator=g, device=dev)[:topk] for _ in range(B)]).contiguous()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        _, _, _, sc = hip_ops.refine_forward(bank, q, init, cand, prob, topk, 1.6, 1000.0, return_scratch=True)
        e1.record()
        torch.cuda.synchronize()
        if it >= 2:
            times.append(e0.elapsed_time(e1)); rows = float(sc[..., 3].sum())
    t = sum(times) / len(times) * 1e-3
    print(f"B = {B:5d}: {rows * 4096 / 1e6:8.1f} MB streamed in {t * 1e6:7.1f} us = {rows * 4096 / t / 1e12:.2f} TB/s ({rows * 4096 / t / 8e12:.2f} of 8 TB/s)")
