"""The deferred exact tier (round 6): `model(**data)` -> `refiner(...)` with the reference's discrete outputs, WITHOUT a host
synchronisation per step and without a latency-bound exact pass per step.

The reference computes the path in fp32 (models/super_guessr.py:447-459, models/proto_refiner.py:176-222).  This path's fast encoder
carries the rounding of 16-bit MFMA operands; a sample whose discrete outputs that error could change (pg_head_certainty,
pg_refine_certainty; error model: pigeon_amd/certainty.py) is re-encoded by the exact encoder.  Round 5 did that inside every step:
`torch.nonzero` (a host synchronisation), then ~9 images through a 24-layer pass whose kernels cannot fill 256 CUs -- 15.7 ms per step
for 1.8 % of the work.  Here

  * every step writes its results into a RING of result slabs (R steps x world x B rows, the layout of the gathered batch);
  * the rows that are not certain are compacted ON THE DEVICE into a circular queue: their pixels and their ring row
    (pg_requeue_append + pg_rows_to_slots); the queue's running count reaches the host ONE STEP LATE through pinned memory and an event
    the GPU passed a whole step ago -- the launch thread never waits for the step it has just queued;
  * when the longest queue of any rank holds `min_flush` panoramas (or a step has waited `max_lag` steps, or at `flush()`), the exact
    tier runs ONCE over queued rows and its results are scattered into the ring (pg_scatter_rows), judged again at the exact tier's
    floor.  A pass takes a whole number of `pass_quantum`s from the head of the queue and leaves the rest for the next one: the exact
    encoder's GEMMs work in 256-row panels, 64 of which (x 4 column tiles at N = 1024) are one round of the 256 CUs -- 16 384 rows = 28
    images = 7 panoramas.  (Several ranks: the pass is triggered by the MEDIAN queue, see `_pass_size`.)  Measured (profiles/r06/exact_sweep.txt): 28 / 56 / 84 / 112 images cost 1.19-1.21 ms per image, 32-44
    images 1.32-1.40, 60 images 1.33: a pass of "whatever is queued" (10-12 panoramas) pays for rounds it leaves half empty;
  * a step is handed out only when all its rows are settled.  The reference's loops collect at the end
    (training/train_eval_loop.py:98-112, preprocessing/embed.py:36-43): handing results out a few steps late changes nothing for them.

Data-parallel runs: the queue counts ride in the step's second all-gather, so every rank takes the SAME flush decision in the SAME step
from the same numbers, and every rank runs the exact tier on the same number of slots (the longest queue; shorter queues pad with
rows that are discarded): equal work per rank whatever each rank found uncertain.  The patched rows travel in one more grouped
all-gather per flush.

The refinement runs ONCE per row: `ProtoRefiner.forward_certain` is the refinement (pg_refine_forward_ex is pg_refine_forward with
records) and its result is kept for every row that was not re-encoded.

`ops` is the module that provides requeue_append / rows_to_slots / requeue_take / scatter_rows: pigeon_amd.hip_ops (the C ABI) -- the
CPU tests and bench.py --dry-run pass torch stand-ins, which are test infrastructure and never reachable from the product.
"""
from __future__ import annotations

from collections import deque
from typing import Dict, List, Optional

import torch

from .utils import TopK


class LocalComm:
    """world_size 1 without any process group: what `certain_forward` / `evaluate_model` (one process, as the reference's evaluate) use."""
    rank, world_size = 0, 1

    def gather_many(self, tensors, out=None):
        if out is None:
            return list(tensors)
        for t, o in zip(tensors, out):
            o.copy_(t)
        return list(out)


class DeferredExact:
    def __init__(self, model, refiner=None, comm=None, ops=None, min_flush: Optional[int] = None, max_lag: int = 12,
                 immediate: bool = False, keep_logits: bool = True, pass_quantum: Optional[int] = None):
        """min_flush: queued panoramas (on the rank with the longest queue) that trigger an exact pass (None: one quantum, or 10 where
        there is no quantum); max_lag: steps a queued row may wait; immediate: settle every step before `submit` returns (one host
        synchronisation per step: the serving / single-call form).  keep_logits: carry the (B, C) logits through the ring (the loss
        `package` computes needs them).  pass_quantum: rows a min_flush pass takes are a multiple of it (0: everything queued; None:
        what fills one round of the device's CUs with the exact encoder's 256-row panels -- `round_quantum`, known at the first
        step with pixels)."""
        if ops is None:
            from . import hip_ops as ops
        self.model, self.refiner, self.ops = model, refiner, ops
        self.comm = comm if comm is not None else LocalComm()
        self._min_flush_arg, self.max_lag, self.immediate = min_flush, int(max_lag), bool(immediate)
        self.pass_quantum = pass_quantum if pass_quantum is None else int(pass_quantum)
        self.min_flush = int(min_flush) if min_flush is not None else (self.pass_quantum or 10)
        self.keep_logits = bool(keep_logits)
        self.R = (2 if immediate else self.max_lag + 4)
        self.ring: Optional[Dict[str, torch.Tensor]] = None
        self.local: Optional[Dict[str, torch.Tensor]] = None
        self.pending: deque = deque()
        self.n_submitted = 0
        self.flush_log: List[dict] = []           # one entry per exact pass: step, per-rank queue lengths, slots run
        self.dropped_checked = 0
        self.boundary_checked = None
        self.marks = None                         # set to a list: submit() appends [start, before gather 1, after, before gather 2, end]
        self.dev = None                           # the model's device, known at the first submit

    # ------------------------------------------------------------------------------------------------ storage
    def _alloc(self, st: dict, B: int, index_dtype=torch.int64):
        dev = st['tol'].device
        W = self.comm.world_size
        self.B, self.W, self.WB, self.dev = B, W, W * B, dev
        R, n = self.R, self.R * W * B
        emb = st['embedding']
        ring = {'embedding': torch.zeros((n,) + tuple(emb.shape[1:]), dtype=emb.dtype, device=dev)}
        for k in ('topk_values', 'topk_indices', 'preds_LLH', 'preds_geocell'):
            ring[k] = torch.zeros((n,) + tuple(st[k].shape[1:]), dtype=st[k].dtype, device=dev)
        ring['index'] = torch.zeros((n,), dtype=index_dtype, device=dev)
        if self.refiner is not None:
            ring['refined_LLH'] = torch.zeros((n, 2), dtype=torch.float32, device=dev)
            ring['refined_geocell'] = torch.zeros((n,), dtype=torch.int64, device=dev)
        ring['certain'] = torch.zeros((n,), dtype=torch.bool, device=dev)       # one byte per row, 0 / 1: the kernels write it as uint8
        ring['cause'] = torch.zeros((n,), dtype=torch.int32, device=dev)
        ring['exact'] = torch.zeros((n,), dtype=torch.bool, device=dev)
        self.ring = ring
        loc = {}                                      # this rank's own reports: never gathered, patched through the remap
        for k in ('tol', 'margin', 'sens') + (('logits',) if self.keep_logits and 'logits' in st else ()):
            if k in st and st[k] is not None:
                loc[k] = torch.zeros((R * B,) + tuple(st[k].shape[1:]), dtype=st[k].dtype, device=dev)
        if self.refiner is not None:
            loc['refine_tol'] = torch.zeros((R * B,), dtype=torch.float32, device=dev)
            loc['refine_code'] = torch.zeros((R * B,), dtype=torch.int32, device=dev)
        self.local = loc
        self._false = torch.zeros((B,), dtype=torch.bool, device=dev)
        self._true = torch.ones((B,), dtype=torch.bool, device=dev)
        self._arange = torch.arange(B, device=dev, dtype=torch.int64)
        self._prev_appended = [0] * W
        # the queue (only when rows can be fixed: pixels + an exact encoder)
        self.cap = 0
        self.q_pixels = self.slot_dst = None
        self.counters = torch.zeros((2,), dtype=torch.int64, device=dev)
        self.counts_all = torch.zeros((R, W), dtype=torch.int64, device=dev)
        self.host_counts = torch.zeros((R, W), dtype=torch.int64)
        if dev.type == 'cuda':
            self.host_counts = self.host_counts.pin_memory()
            self.events = [torch.cuda.Event() for _ in range(R)]
        else:
            self.events = [None] * R
        self.flushed = [0] * W
        self.px_shape = None

    def _alloc_queue(self, px_rows: torch.Tensor):
        if self.pass_quantum is None:
            self.pass_quantum = round_quantum(px_rows)
            if self._min_flush_arg is None:
                self.min_flush = self.pass_quantum or 10
        self.cap = (self.B if self.immediate else self.min_flush + (self.pass_quantum or 0) + 2 * self.B)
        self.q_pixels = torch.zeros((self.cap,) + tuple(px_rows.shape[1:]), dtype=px_rows.dtype, device=px_rows.device)
        self.slot_dst = torch.zeros((self.cap,), dtype=torch.int64, device=px_rows.device)
        self._true_cap = torch.ones((self.cap,), dtype=torch.bool, device=px_rows.device)
        self.px_shape = (tuple(px_rows.shape[1:]), px_rows.dtype)

    def _mark(self, m):
        if m is None:
            return
        if self.dev.type == 'cuda':
            ev = torch.cuda.Event(enable_timing=True)
            ev.record()
            m.append(ev)
        else:
            import time
            m.append(time.perf_counter())

    # ------------------------------------------------------------------------------------------------ one step
    @torch.no_grad()
    def submit(self, pixel_values=None, embedding=None, index=None, meta=None) -> List[dict]:
        """Queue one step.  Returns the results that became final (possibly none, possibly several), oldest first; each is a dict of
        the gathered batch's tensors (rank-major; copies, valid for as long as the caller keeps them): embedding, index, preds_geocell,
        preds_LLH, topk_indices, topk_values [, refined_LLH, refined_geocell], certain (bool), cause, exact (bool: re-encoded rows),
        plus `step`, `meta` (what the caller passed), `state` (this rank's rows in the form `SuperGuessr.package` takes)."""
        model, comm, ops = self.model, self.comm, self.ops
        marks = [] if self.marks is not None else None
        settled = []
        if self.dev is None:
            self.dev = model.cell_layer.weight.device
        self._mark(marks)
        st = model.encode_head(pixel_values, embedding)
        b = int(st['tol'].shape[0])
        px = st.get('pixel_values')
        if b == 0:
            # an empty batch (a rank whose shard is empty must still be able to step): nothing to queue, nothing to patch -- it takes its
            # place in the order of hand-out and carries the head's (empty) outputs as they are
            st['pixel_values'] = None
            empty = self._empty_result(st, index, meta)
            if not self.pending:
                return [empty]
            self.pending.append(dict(step=self.n_submitted, slot=-1, b=0, meta=meta, appended=None, can_fix=False, ready=empty))
            self.n_submitted += 1
            return self._advance(final=False)
        can_fix = bool(getattr(model, 'exact_top1', False)) and px is not None and not st.get('exact_tier', False)
        if self.ring is not None and (b > self.B or (comm.world_size > 1 and b != self.B)):
            out = self.flush()                                   # another batch geometry: settle what is pending, start over
            self.ring = None
            st['pixel_values'] = None
            return out + self.submit(pixel_values, embedding, index, meta)
        if self.ring is None:
            self._alloc(st, b)
        rank, B, WB = comm.rank, self.B, self.WB
        step = self.n_submitted
        slot = step % self.R
        if any(r['slot'] == slot for r in self.pending):         # the ring has come round to a slab that is still pending
            raise RuntimeError('DeferredExact: result ring overrun (max_lag too large for the ring)')
        gbase, lbase = slot * WB, slot * B
        dev = self.dev
        if index is None:
            index = self._arange[:b] if rank == 0 else self._arange[:b] + rank * b
        index = index.to(dev)
        exact_tier = bool(st.get('exact_tier', False))
        thr = st['thr']
        for k, buf in self.local.items():
            if k in st:
                buf[lbase:lbase + b].copy_(st[k])
        ring = self.ring
        slab = lambda k: ring[k][gbase:gbase + comm.world_size * b]            # noqa: E731  (world 1: b <= B rows of the slab)
        own = slice(gbase + rank * b, gbase + (rank + 1) * b)
        self._mark(marks)
        # the gather BEFORE the refinement: `north_star` / the reference's accelerator.gather (preprocessing/embed.py:36-37)
        comm.gather_many([st['embedding'], st['topk_indices'], st['topk_values'], st['preds_LLH'], st['preds_geocell'], index],
                         out=[slab('embedding'), slab('topk_indices'), slab('topk_values'), slab('preds_LLH'), slab('preds_geocell'),
                              slab('index')])
        self._mark(marks)
        rtol = rcode = None
        W_head = model.cell_layer.weight.data
        if self.refiner is not None:
            # THE refinement of this rank's slice (pg_refine_forward_ex = pg_refine_forward + the records the certainty pass reads)
            llh_r, cell_r, rtol, rcode, self.boundary_checked = self.refiner.forward_certain(
                ring['embedding'][own], ring['preds_LLH'][own], ring['topk_indices'][own], ring['topk_values'][own], W_head,
                st['wstats'], st.get('drift'))
            self.local['refine_tol'][lbase:lbase + b].copy_(rtol)
            self.local['refine_code'][lbase:lbase + b].copy_(rcode)
        wants_queue = bool(getattr(model, 'exact_top1', False)) and px is not None
        if wants_queue:
            # (also a rank whose whole batch went through the exact encoder -- `exact_tier`: it queues nothing, but in a data-parallel job
            # it still runs the padded exact passes the other ranks' queues ask for)
            px_rows = px.reshape((b, -1)).contiguous()            # (a strided view handed in by the caller: one copy; else free)
            if self.q_pixels is None or self.px_shape != (tuple(px_rows.shape[1:]), px_rows.dtype):
                if self.q_pixels is not None:
                    # another pixel dtype / geometry (e.g. fp16 pixels from the GPU preprocessing after fp32 tensors): settle what is
                    # queued in the old one, then start a new queue (every rank sees the same change in the same step)
                    settled = self.flush()
                    self.counters.zero_()
                    self.flushed = [0] * comm.world_size
                    self._prev_appended = [0] * comm.world_size
                self._alloc_queue(px_rows)
        if can_fix:
            certain, cause, row_slot = ops.requeue_append(st['tol'], rtol, rcode, thr, False, dst_base=gbase + rank * b,
                                                          flushed=self.flushed[rank], cap=self.cap, counters=self.counters,
                                                          slot_dst=self.slot_dst)
            ops.rows_to_slots(px_rows, row_slot, self.q_pixels)
        else:
            certain, cause, _ = ops.requeue_append(st['tol'], rtol, rcode, thr, False)
        small = [certain.view(torch.bool), cause, (self._true if exact_tier else self._false)[:b], self.counters[:1]]
        outs = [slab('certain'), slab('cause'), slab('exact'), self.counts_all[slot]]
        if self.refiner is not None:
            small += [llh_r, cell_r]
            outs += [slab('refined_LLH'), slab('refined_geocell')]
        self._mark(marks)
        comm.gather_many(small, out=outs)
        self.host_counts[slot].copy_(self.counts_all[slot], non_blocking=True)
        if self.events[slot] is not None:
            self.events[slot].record()
        self._mark(marks)
        if marks is not None:
            self.marks.append(marks)
        self.pending.append(dict(step=step, slot=slot, b=b, meta=meta, appended=None, can_fix=can_fix))
        self.n_submitted += 1
        st['pixel_values'] = None                                # the queue holds what it needs; do not keep a batch of pixels alive
        return settled + self._advance(final=False)

    def flush(self) -> List[dict]:
        """Settle everything that is pending (one host synchronisation) and hand it out."""
        return self._advance(final=True)

    # ------------------------------------------------------------------------------------------------ host side of the queue
    def _empty_result(self, st: dict, index, meta) -> dict:
        dev = st['tol'].device
        k = int(getattr(self.model, 'num_candidates', st['topk_indices'].shape[1]))
        z = lambda dt, *shape: torch.zeros((0,) + shape, dtype=dt, device=dev)     # noqa: E731
        state = {kk: st[kk] for kk in ('embedding', 'topk_values', 'topk_indices', 'preds_LLH', 'preds_geocell', 'tol', 'margin', 'sens',
                                       'logits') if kk in st}
        state.update(certain=z(torch.bool), exact=z(torch.bool), cause=z(torch.int32), pixel_values=None)
        res = dict(embedding=st['embedding'], topk_indices=st['topk_indices'][:, :k], topk_values=st['topk_values'][:, :k],
                   preds_LLH=st['preds_LLH'], preds_geocell=st['preds_geocell'], index=z(torch.int64), certain=z(torch.bool),
                   exact=z(torch.bool), cause=z(torch.int32), queued=[0] * self.comm.world_size, step=self.n_submitted, meta=meta, state=state)
        if self.refiner is not None:
            res['refined_LLH'], res['refined_geocell'] = z(torch.float32, 2), z(torch.int64)
            state.update(refined_LLH=res['refined_LLH'], refined_geocell=res['refined_geocell'], refine_tol=z(torch.float32),
                         refine_code=z(torch.int32))
        return res

    def _learn(self, upto_step: int):
        last_known = None
        for rec in self.pending:
            if 'ready' in rec:                                   # an empty batch: queues nothing, complete as soon as its turn comes
                if rec['appended'] is None and (last_known is not None or rec is self.pending[0]):
                    rec['appended'] = list(last_known) if last_known is not None else list(self.flushed)
                last_known = rec['appended'] if rec['appended'] is not None else last_known
                continue
            if rec['appended'] is not None:
                last_known = rec['appended']
            if rec['step'] <= upto_step and rec['appended'] is None:
                ev = self.events[rec['slot']]
                if ev is not None:
                    ev.synchronize()
                rec['appended'] = [int(v) for v in self.host_counts[rec['slot']].tolist()]
                last_known = rec['appended']

    def _advance(self, final: bool) -> List[dict]:
        if not self.pending:
            return []
        last = self.n_submitted - 1
        # counts the GPU produced a whole step ago: waiting for them never starves it (the step just queued is still to run)
        self._learn(last if (final or self.immediate) else last - 1)
        known = [r for r in self.pending if r['appended'] is not None]
        if known:
            app = known[-1]['appended']
            lens = [a - f for a, f in zip(app, self.flushed)]
            if max(lens) > 0:
                first_waiting = next((r for r in known if any(a > f for a, f in zip(r['appended'], self.flushed))), None)
                waited = last - first_waiting['step'] if first_waiting is not None else 0
                if final or self.immediate or waited >= self.max_lag:
                    self._exact_pass(lens, at_step=last)
                else:
                    take = self._pass_size(lens)
                    if take > 0:
                        self._exact_pass([min(n, take) for n in lens], at_step=last)   # the head of every queue; the rest waits
        done = []
        while self.pending and self.pending[0]['appended'] is not None and \
                all(f >= a for a, f in zip(self.pending[0]['appended'], self.flushed)):
            done.append(self._emit(self.pending.popleft()))
        return done

    def _pass_size(self, lens: List[int]) -> int:
        """Slots the next regular pass takes from the head of every queue (0: no pass yet) -- a pure function of the gathered queue
        lengths, so every rank decides the same.  Without a quantum: everything, once the longest queue holds min_flush.  With one: a
        whole number of quanta, when the MEDIAN queue (lower median; one rank: the queue) holds min_flush -- every rank runs the same
        number of slots, so a pass triggered by the longest of 8 queues while the others hold 4-5 rows runs them a third empty
        (tools/pass_policy_sim.py, Poisson arrivals: 1.54 slots run per real row on 8 ranks, 1.36 on 4, 1.18 on 2); at the median half
        the ranks fill their slots and the others miss one or two (1.17 / 1.12 / 1.05) -- or when the longest queue has got one
        quantum ahead of min_flush (a rank that finds more: the pass then takes all but about one quantum of it), so that no queue
        holds more than min_flush + quantum + the two steps the host has not seen."""
        q = self.pass_quantum or 0
        longest = max(lens)
        if not q or longest < q:
            return longest if longest >= self.min_flush else 0
        median = sorted(lens)[(len(lens) - 1) // 2]
        take = 0
        if median >= self.min_flush:
            take = q * max(1, median // q)
        if longest >= self.min_flush + q:
            take = max(take, q * max(1, (longest - q) // q))
        return take

    @torch.no_grad()
    def _exact_pass(self, lens: List[int], at_step: int):
        """One exact pass over the first lens[r] queued rows of every rank r (every rank runs max(lens) slots)."""
        model, comm, ops = self.model, self.comm, self.ops
        rank, W = comm.rank, comm.world_size
        n_own, n_pad = lens[rank], max(lens)
        head = self.flushed[rank]
        span = [] if self.marks is not None else None
        self._mark(span)
        if self.cap <= 0:
            raise RuntimeError('DeferredExact: rows are queued but there is no queue (internal error)')
        h0 = head % self.cap
        seg = [self.q_pixels[h0:min(self.cap, h0 + n_pad)]]
        if h0 + n_pad > self.cap:
            seg.append(self.q_pixels[:h0 + n_pad - self.cap])
        sx = model.exact_rows(seg)                                # exact encoder + head + tolerance at the exact tier's floor
        thr_x = model.certainty.threshold(exact=True)
        rtol = rcode = None
        if self.refiner is not None:
            llh_x, cell_x, rtol, rcode, _ = self.refiner.forward_certain(sx['embedding'], sx['preds_LLH'], sx['topk_indices'],
                                                                         sx['topk_values'], model.cell_layer.weight.data,
                                                                         model.wstats(True), None)
            sx['refine_tol'], sx['refine_code'] = rtol, rcode
        certain_x, _, _ = ops.requeue_append(sx['tol'], rtol, rcode, thr_x, False)
        dst = ops.requeue_take(self.slot_dst, head, n_own, n_pad)
        cols = [('embedding', sx['embedding']), ('topk_indices', sx['topk_indices']), ('topk_values', sx['topk_values']),
                ('preds_LLH', sx['preds_LLH']), ('preds_geocell', sx['preds_geocell']), ('certain', certain_x.view(torch.bool)),
                ('exact', self._true_cap[:n_pad])]
        if self.refiner is not None:
            cols += [('refined_LLH', llh_x), ('refined_geocell', cell_x)]
        srcs = [c for _, c in cols] + [dst]
        if W > 1 or getattr(comm, 'force_rccl', False):
            srcs = comm.gather_many(srcs)                         # every rank patches every rank's rows of its copy of the ring
        dst_all = srcs[-1]
        for (k, _), src in zip(cols, srcs[:-1]):
            ops.scatter_rows(src, dst_all, self.ring[k])
        for k, buf in self.local.items():                         # this rank's own reports (margin, tolerances, logits)
            if k in sx:
                ops.scatter_rows(sx[k], dst, buf, remap=(self.WB, self.B, rank * self.B))
        for r in range(W):
            self.flushed[r] += lens[r]
        self._mark(span)
        self.flush_log.append(dict(at_step=at_step, queued=list(lens), slots_run=n_pad, span=span))

    def exact_pass_ms(self, entries=None) -> List[float]:
        """Milliseconds of the exact passes logged while `marks` was on (stream time stamps; synchronise the device first)."""
        out = []
        for f in (self.flush_log if entries is None else entries):
            sp = f.get('span')
            if sp:
                out.append((sp[1] - sp[0]) * 1e3 if isinstance(sp[0], float) else sp[0].elapsed_time(sp[1]))
        return out

    def _emit(self, rec: dict) -> dict:
        if 'ready' in rec:
            return rec['ready']
        comm = self.comm
        rank, b = comm.rank, rec['b']
        gbase, lbase = rec['slot'] * self.WB, rec['slot'] * self.B
        n = comm.world_size * b
        ring = self.ring
        res = {k: ring[k][gbase:gbase + n].clone() for k in ring}
        k = int(getattr(self.model, 'num_candidates', res['topk_indices'].shape[1]))
        own = slice(rank * b, (rank + 1) * b)
        state = {'embedding': res['embedding'][own], 'topk_values': res['topk_values'][own], 'topk_indices': res['topk_indices'][own],
                 'preds_LLH': res['preds_LLH'][own], 'preds_geocell': res['preds_geocell'][own],
                 'certain': res['certain'][own], 'exact': res['exact'][own], 'cause': res['cause'][own], 'pixel_values': None}
        if self.refiner is not None:
            state['refined_LLH'], state['refined_geocell'] = res['refined_LLH'][own], res['refined_geocell'][own]
        for kk, buf in self.local.items():
            state[kk] = buf[lbase:lbase + b].clone()
        res['topk_indices'] = res['topk_indices'][:, :k]
        res['topk_values'] = res['topk_values'][:, :k]
        prev = self._prev_appended
        res['queued'] = [a - p for a, p in zip(rec['appended'], prev)]      # rows each rank sent to the exact tier in this step
        self._prev_appended = rec['appended']
        res['step'], res['meta'], res['state'] = rec['step'], rec['meta'], state
        return res

    def check_nothing_dropped(self) -> int:
        """Rows that did not fit the queue since creation (host synchronisation).  The queue is sized so that this is 0."""
        return int(self.counters[1].item()) if self.ring is not None else 0


def round_quantum(px_rows: torch.Tensor, tokens: int = 577, panel_rows: int = 256, n_tiles: int = 4) -> int:
    """Queue rows (panoramas, or single images) whose token rows fill ONE round of the device's CUs with the exact encoder's 256-row
    panels at N = 1024 (pg_vit_forward_precise runs the 256 x 256 persistent kernel: 4 column tiles per panel): 256 CUs -> 64 panels =
    16 384 token rows = 28 images = 7 four-view panoramas.  0 (no quantum) off the GPU."""
    if px_rows.device.type != 'cuda':
        return 0
    cus = torch.cuda.get_device_properties(px_rows.device).multi_processor_count
    images = (cus // n_tiles) * panel_rows // tokens
    views = max(1, int(px_rows.shape[1]) // (3 * 336 * 336))
    return max(1, images // views)


def as_model_output(model, res: dict, labels=None, labels_clf=None):
    """One emitted result -> what `SuperGuessr.forward` returns for this rank's rows (ModelOutput or the serving tuple)."""
    return model.package(dict(res['state']), labels, labels_clf)


__all__ = ['DeferredExact', 'LocalComm', 'as_model_output', 'round_quantum', 'TopK']
