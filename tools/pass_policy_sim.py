"""Round 6: what the exact tier's padding costs in a data-parallel job, by policy (host arithmetic only, no GPU).  W ranks, every rank
finds Poisson(rate) panoramas per step uncertain; all ranks run every exact pass on the same number of slots (the padded ones are
discarded), so what matters is slots run per real row.  Policies: 'longest' = a pass when the LONGEST queue holds one quantum (the
single-rank rule applied to the longest queue), 'median' = DeferredExact._pass_size (a pass when the lower-median queue holds one quantum,
or the longest has got a quantum ahead).   python tools/pass_policy_sim.py [ranks] [rate] [steps]"""
import sys

import numpy as np

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from pigeon_amd.deferred import DeferredExact  # noqa: E402


def simulate(policy, W, rate, steps, q=7, seed=0):
    rng = np.random.default_rng(seed)
    eng = DeferredExact.__new__(DeferredExact)
    eng.pass_quantum, eng.min_flush = q, q
    lens = [0] * W
    slots = rows = passes = 0
    longest_seen = 0
    for _ in range(steps):
        arrivals = rng.poisson(rate, W)
        known = list(lens)                                   # the host decides from the counts of the step before
        lens = [a + b for a, b in zip(lens, arrivals)]
        if policy == "median":
            take = eng._pass_size(known)
        else:
            take = (max(known) // q) * q if max(known) >= q else 0
        if take > 0:
            took = [min(n, take) for n in known]
            slots += max(took) * W
            rows += sum(took)
            passes += 1
            lens = [n - t for n, t in zip(lens, took)]
        longest_seen = max(longest_seen, max(lens))
    return slots / max(rows, 1), passes / steps, longest_seen


if __name__ == "__main__":
    W = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    rate = float(sys.argv[2]) if len(sys.argv) > 2 else 2.25
    steps = int(sys.argv[3]) if len(sys.argv) > 3 else 20000
    for Wn in sorted({1, 2, 4, W}):
        for policy in ("longest", "median"):
            r, f, m = simulate(policy, Wn, rate, steps)
            print(f"{Wn} ranks, {rate} uncertain panoramas per rank and step, policy {policy:8s}: {r:.3f} slots run per real row, "
                  f"{f:.3f} passes per step, longest queue ever {m}")
