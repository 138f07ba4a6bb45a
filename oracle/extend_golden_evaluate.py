"""Add evaluate()'s refinement (topk 40 of 50, T 0.6, 100000 km; reference evaluation/evaluate.py:44,79-80) to the two
128-panorama fixtures tests/golden/pipeline24_wide.npz and pipeline24_spread.npz.

TEST INFRASTRUCTURE, authoring container only (needs /root/reference).  The fixtures already hold the REAL reference's
embeddings and its 50 candidates per panorama (oracle/make_golden.py); this script feeds exactly those to the REAL reference
ProtoRefiner (models/proto_refiner.py:121-231) at evaluate()'s parameters and appends `evaluate_LLH`, `evaluate_cell`,
`evaluate_params` to the same .npz (every other array is left as it is, bit for bit).  The prototype bank is the one the
fixture was made with: synthetic.make_bank(C, 4, seed 2, centre / radius stored in the fixture).

    python oracle/extend_golden_evaluate.py [--only pipeline24_wide|pipeline24_spread]
"""
import argparse
import os
import sys
import tempfile
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from oracle import reference_loader  # noqa: E402
from pigeon_amd import synthetic  # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")


def extend(name: str, tmp: str):
    path = os.path.join(GOLD, name + ".npz")
    z = dict(np.load(path))
    C = int(z["meta"][4])
    geo_csv = os.path.join(tmp, "geocells10k.csv")
    synthetic.write_geocell_csv(geo_csv, synthetic.make_geocells(C, seed=0))
    bank = synthetic.make_bank(C, int(z["meta"][5]), seed=int(z["meta"][6]), empty_frac=0.01, max_members=int(z["meta"][7]),
                               center=z["center"], radius=float(z["radius"]))
    proto_csv, ds_dir = os.path.join(tmp, name + "_protos.csv"), os.path.join(tmp, name + "_train")
    synthetic.write_bank_reference_files(bank, proto_csv, ds_dir)
    ns = reference_loader.load(geo_csv, proto_csv, ds_dir)
    topk, T, mr = 40, 0.6, 100000
    ref = ns.ProtoRefiner(topk=topk, max_refinement=mr, temperature=T, proto_path=proto_csv, dataset_path=ds_dir, protos=[None] * C)
    import datasets as _ds
    _ds.disable_progress_bar()
    cand = torch.from_numpy(z["topk_indices"])
    needed = sorted(set(cand[:, :topk].flatten().tolist()))
    built = [None] * C
    t0 = time.time()
    for n_done, c in enumerate(needed):
        built[c] = ref._get_prototypes(c)
        if n_done % 250 == 0:
            print(f"{name}: prototypes of {n_done}/{len(needed)} candidate cells, {time.time() - t0:.0f} s", flush=True)
    ref.protos = built
    ref.eval()
    emb = torch.from_numpy(z["embedding"])
    with torch.no_grad():
        # the class defaults again on the way: must reproduce what the fixture already holds (the bank really is the same one)
        ref.topk, ref.max_refinement = 5, 1000
        ref.temperature.data = torch.tensor(1.6)
        _, llh5, cell5 = ref(emb, initial_preds=torch.from_numpy(z["preds_LLH"]), candidate_cells=cand,
                             candidate_probs=torch.from_numpy(z["topk_values"]))
        assert np.array_equal(cell5.numpy(), z["default_cell"]) and np.array_equal(llh5.numpy(), z["default_LLH"]), \
            "the rebuilt bank does not reproduce the fixture's default refinement"
        ref.topk, ref.max_refinement = topk, mr
        ref.temperature.data = torch.tensor(T)
        _, llh, cell = ref(emb, initial_preds=torch.from_numpy(z["preds_LLH"]), candidate_cells=cand,
                           candidate_probs=torch.from_numpy(z["topk_values"]))
    z["evaluate_LLH"], z["evaluate_cell"] = llh.numpy(), cell.numpy()
    z["evaluate_params"] = np.array([topk, T, mr], dtype=np.float64)
    print(f"{name}: evaluate() refinement changed", int((cell.numpy() != z["preds_geocell"]).sum()), "of", len(cell), flush=True)
    np.savez(path, **z)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--only", default=None)
    args = ap.parse_args()
    torch.set_num_threads(max(1, (os.cpu_count() or 2) // 2))
    tmp = tempfile.mkdtemp(prefix="pigeon_golden_eval_")
    for name in ("pipeline24_wide", "pipeline24_spread"):
        if args.only in (None, name):
            extend(name, tmp)


if __name__ == "__main__":
    main()
