"""ORACLE -- host restatement of csrc/requeue.hip (the device side of the deferred exact tier) in plain torch.
TEST INFRASTRUCTURE, NOT PRODUCT: only tests/ and bench.py --dry-run import this (pigeon_amd/ never does).

This is bookkeeping of THIS implementation, not arithmetic of the reference: the reference has no exact tier because it is fp32
end to end (models/super_guessr.py:447-459, models/proto_refiner.py:176-222).  The functions restate include/pigeon_hip.h's contract
for pg_requeue_append / pg_rows_to_slots / pg_requeue_take / pg_scatter_rows / pg_head_wstats entry by entry, so that (a) the GPU
parity tests can compare the kernels with something written independently of them, and (b) the host logic of
pigeon_amd.deferred.DeferredExact can be exercised on CPU tensors (world_size 1 and 2, gloo) with these as its `ops`.
"""
import torch


def requeue_append(head_tol, refine_tol, refine_code, thr, force_all=False, dst_base=0, flushed=0, cap=0, counters=None, slot_dst=None):
    B = head_tol.numel()
    thr = float(torch.tensor(thr, dtype=torch.float32))            # the kernel compares in fp32
    h_unc = torch.ones(B, dtype=torch.bool) if force_all else ~(head_tol > thr)
    r_unc = ~(refine_tol > thr) if refine_tol is not None else torch.zeros(B, dtype=torch.bool)
    h_unc, r_unc = h_unc.to(head_tol.device), r_unc.to(head_tol.device)
    flag = h_unc | r_unc
    cause = torch.zeros(B, dtype=torch.int32, device=head_tol.device)
    if refine_tol is not None:
        code = refine_code if refine_code is not None else torch.full((B,), 2, dtype=torch.int32, device=head_tol.device)
        code = torch.where(code == 0, torch.full_like(code, 2), code)
        cause = torch.where(r_unc, code.to(torch.int32), cause)
    cause = torch.where(h_unc, torch.ones_like(cause), cause)
    certain = (~flag).to(torch.uint8)
    row_slot = None
    if cap > 0:
        row_slot = torch.full((B,), -1, dtype=torch.int32, device=head_tol.device)
        appended, dropped = int(counters[0]), int(counters[1])
        for r in torch.nonzero(flag).flatten().tolist():          # row order
            if appended - flushed < cap:
                slot = appended % cap
                slot_dst[slot] = dst_base + r
                row_slot[r] = slot
                appended += 1
            else:
                row_slot[r] = -2
                dropped += 1
        counters[0], counters[1] = appended, dropped
    return certain, cause, row_slot


def rows_to_slots(src, row_slot, dst):
    for r, s in enumerate(row_slot.tolist()):
        if s >= 0:
            dst[s] = src[r]


def requeue_take(slot_dst, head, n_valid, n_pad):
    cap = slot_dst.numel()
    out = torch.full((n_pad,), -1, dtype=torch.int64, device=slot_dst.device)
    for i in range(n_valid):
        out[i] = slot_dst[(head + i) % cap]
    return out


def scatter_rows(src, dst_row, dst, remap=None):
    for i, d in enumerate(dst_row.tolist()):
        if d < 0:
            continue
        if remap is not None:
            wb, b, off = remap
            slab, r = d // wb, d % wb - off
            if r < 0 or r >= b:
                continue
            d = slab * b + r
        if d >= dst.shape[0]:
            continue
        dst[d] = src[i]


def head_wstats(W, drift):
    Wf = W.float()
    wb = (Wf @ drift.float()).abs().max() if drift is not None else torch.zeros((), device=W.device)
    return torch.stack([Wf.norm(dim=1).max(), wb.float()])
