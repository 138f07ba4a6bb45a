"""Host-side logic of the kept entry points, on CPU (no GPU, no compute calls into the library):

  SuperGuessr.load_state / load_state_dict      reference models/super_guessr.py:222-238, models/utils.py:24-45
  bank_from_protos / HostBank.save|load          reference evaluation/evaluate.py:66-75 (pickled `refiner.protos`)
  compute_geoguessr_metrics                      reference evaluation/metrics.py:89-181
  shard_batches / compute_embeddings on disk     reference preprocessing/embed.py:16-43,68, dataset_preprocessing.py:294-300
  evaluate() argument handling                   reference evaluation/evaluate.py:42-47
Where the reference tree is present (/root/reference, authoring container) the inputs come from the reference's own
classes through oracle/reference_loader.py; the committed fixtures (tests/golden/geo.npz) cover the rest.
"""
import os
import sys

import numpy as np
import pytest
import torch

from oracle import reference_loader
from pigeon_amd import synthetic

needs_reference = pytest.mark.skipif(not reference_loader.available(), reason="reference tree not present")


def _geo_csv(tmp_path, C=37):
    p = os.path.join(str(tmp_path), "geocells.csv")
    synthetic.write_geocell_csv(p, synthetic.make_geocells(C, seed=0))
    return p


# ------------------------------------------------------------------------------------------------ checkpoints
def _vit_sd(layers=1, seed=3):
    return synthetic.make_vit_weights(seed=seed, layers=layers, affine_jitter=True)


@pytest.mark.parametrize("layout", ["flat", "vision_model"])
def test_super_guessr_load_state_copies_head_and_base(tmp_path, layout, capsys):
    """A full SuperGuessr checkpoint, in the transformers >= 5 (flat) and the 4.23.1 (`base_model.vision_model.*`,
    reference env.yml:60) key layouts, lands in cell_layer AND in the HIP encoder's weights."""
    from pigeon_amd.clip_embedder import HipCLIPVisionModel
    from pigeon_amd.super_guessr import SuperGuessr
    geo = _geo_csv(tmp_path)
    model = SuperGuessr(HipCLIPVisionModel(_vit_sd(seed=3), layers=1), panorama=True, freeze_base=True, geocell_path=geo)
    src = _vit_sd(seed=4)
    g = torch.Generator().manual_seed(0)
    ckpt = {"cell_layer.weight": torch.randn((37, 1024), generator=g), "cell_layer.bias": torch.randn((37,), generator=g),
            "lla_geocells": torch.zeros((37, 2), dtype=torch.float64)}
    pre = "base_model." + ("vision_model." if layout == "vision_model" else "")
    ckpt.update({pre + k: v for k, v in src.items()})
    ckpt["hedge_layer.weight"] = torch.zeros(3)                       # unknown name: skipped with the reference's message
    path = os.path.join(str(tmp_path), "full.model")
    torch.save(ckpt, path)
    model.load_state(path)
    assert "Parameter hedge_layer.weight not in model's state." in capsys.readouterr().out
    assert torch.equal(model.cell_layer.weight.data, ckpt["cell_layer.weight"])
    assert torch.equal(model.cell_layer.bias.data, ckpt["cell_layer.bias"])
    own = model.base_model.state_dict()
    for k, v in src.items():
        assert torch.equal(own[k], v), k
    assert bool((model.lla_geocells.data == 0).all())


def test_load_state_raises_when_nothing_matches(tmp_path):
    from pigeon_amd.super_guessr import SuperGuessr
    model = SuperGuessr(None, panorama=True, geocell_path=_geo_csv(tmp_path))
    path = os.path.join(str(tmp_path), "bad.model")
    torch.save({"totally.unrelated": torch.zeros(2)}, path)
    with pytest.raises(KeyError, match="none of the 1 parameters"):
        model.load_state(path)


@pytest.mark.parametrize("prefix,embedder", [("", False), ("vision_model.", False), ("base_model.", True),
                                             ("base_model.vision_model.", True)])
def test_load_state_dict_layouts(prefix, embedder):
    """models/utils.py:24-45 incl. the embedder=True strip (:34-35) used by CLIPEmbedding (clip_embedder.py:30-32)."""
    from pigeon_amd.clip_embedder import HipCLIPVisionModel
    from pigeon_amd.utils import load_state_dict
    m = HipCLIPVisionModel(_vit_sd(seed=3), layers=1)
    src = _vit_sd(seed=5)
    n = load_state_dict(m, {prefix + k: v for k, v in src.items()}, embedder=embedder)
    assert n == len(src)
    for k, v in src.items():
        assert torch.equal(m.state_dict()[k], v), k
    with pytest.raises(KeyError):
        load_state_dict(m, {"nope." + k: v for k, v in src.items()})
    # HF-style entry point of the module itself
    m2 = HipCLIPVisionModel(_vit_sd(seed=3), layers=1)
    m2.load_state_dict({"vision_model." + k: v for k, v in src.items()})
    assert torch.equal(m2.state_dict()["encoder.layers.0.mlp.fc1.weight"], src["encoder.layers.0.mlp.fc1.weight"])


@needs_reference
def test_load_state_reads_a_checkpoint_written_by_the_reference(tmp_path):
    """torch.save(reference SuperGuessr(HF CLIPVisionModel).state_dict()) -> our load_state: every tensor arrives."""
    from transformers import CLIPVisionConfig, CLIPVisionModel
    from pigeon_amd.clip_embedder import HipCLIPVisionModel
    from pigeon_amd.super_guessr import SuperGuessr
    geo = _geo_csv(tmp_path)
    ns = reference_loader.load(geo, "unused.csv", "unused_dir")
    cfg = CLIPVisionConfig(hidden_size=1024, intermediate_size=4096, num_hidden_layers=1, num_attention_heads=16,
                           image_size=336, patch_size=14, projection_dim=768)
    torch.manual_seed(7)
    ref_model = ns.SuperGuessr(CLIPVisionModel(cfg), panorama=True, freeze_base=True, num_candidates=5)
    path = os.path.join(str(tmp_path), "ref.model")
    torch.save(ref_model.state_dict(), path)
    ours = SuperGuessr(HipCLIPVisionModel(_vit_sd(seed=3), layers=1), panorama=True, freeze_base=True, geocell_path=geo)
    ours.load_state(path)
    ref_sd = ref_model.state_dict()
    mine = ours.state_dict()
    n_base = 0
    for k, v in ref_sd.items():
        kk = k.replace("base_model.vision_model.", "base_model.")
        if not v.is_floating_point():
            continue
        assert kk in mine, k
        assert torch.equal(mine[kk].cpu(), v), k
        n_base += kk.startswith("base_model.")
    assert n_base >= 20


# ------------------------------------------------------------------------------------------------ prototype bank formats
def _arrays_equal(a, b):
    from pigeon_amd.proto_refiner import HostBank
    for f in HostBank.FIELDS:
        x, y = np.asarray(getattr(a, f)), np.asarray(getattr(b, f))
        assert x.shape == y.shape and np.array_equal(x, y), f


def test_host_bank_save_load_roundtrip(tmp_path):
    from pigeon_amd.proto_refiner import HostBank
    bank = synthetic.make_bank(23, 5, seed=4, empty_frac=0.1)
    hb = HostBank(**{f: getattr(bank, f) for f in HostBank.FIELDS})
    p = os.path.join(str(tmp_path), "proto.refiner.npz")
    hb.save(p)
    back = HostBank.load(p)
    _arrays_equal(hb, back)
    assert back.num_cells == 23 and back.proto_emb.dtype == np.float32 and back.cell_off.dtype == np.int64


@needs_reference
def test_bank_from_protos_on_the_references_own_protos(tmp_path):
    """`refiner.protos` as the REFERENCE builds it (list of per-cell HF Datasets / None, the object
    evaluation/evaluate.py:66-75 pickles) -> bank_from_protos -> exactly the CSR bank the files were written from."""
    from pigeon_amd.proto_refiner import bank_from_protos
    C = 30
    bank = synthetic.make_bank(C, 6, seed=2, empty_frac=0.1)
    proto_csv = os.path.join(str(tmp_path), "protos.csv")
    ds_dir = os.path.join(str(tmp_path), "hf_train")
    synthetic.write_bank_reference_files(bank, proto_csv, ds_dir)
    ns = reference_loader.load(_geo_csv(tmp_path, C), proto_csv, ds_dir)
    ref = ns.ProtoRefiner(topk=5, proto_path=proto_csv, dataset_path=ds_dir)
    assert sum(p is None for p in ref.protos) == int((np.diff(bank.cell_off) == 0).sum()) > 0
    hb = bank_from_protos(ref.protos, ds_dir)
    _arrays_equal(hb, bank)


@needs_reference
def test_load_refiner_cache_reads_the_references_pickle(tmp_path):
    """evaluation/evaluate.py:64-75: the reference caches the whole refiner with `torch.save(refiner, proto_model_path)` and
    reads `torch.load(...).protos` back.  Written here by the reference's OWN class exactly that way; read back without the
    reference package importable (the loader must not need `models.proto_refiner`), converted to the CSR bank."""
    import types
    from pigeon_amd.proto_refiner import bank_from_protos, load_refiner_cache
    C = 24
    bank = synthetic.make_bank(C, 5, seed=3, empty_frac=0.1)
    proto_csv = os.path.join(str(tmp_path), "protos.csv")
    ds_dir = os.path.join(str(tmp_path), "hf_train")
    synthetic.write_bank_reference_files(bank, proto_csv, ds_dir)
    ns = reference_loader.load(_geo_csv(tmp_path, C), proto_csv, ds_dir)
    ref = ns.ProtoRefiner(20, False, 10000, proto_path=proto_csv, dataset_path=ds_dir, temperature=1)     # evaluate.py:73-74
    path = os.path.join(str(tmp_path), "proto.refiner")
    # pickle locates classes by module path: expose the reference's module under its real name for the duration of the save
    mods = {"models": types.ModuleType("models"), "models.proto_refiner": types.ModuleType("models.proto_refiner")}
    mods["models.proto_refiner"].ProtoRefiner = ns.ProtoRefiner
    saved = {k: sys.modules.get(k) for k in mods}
    sys.modules.update(mods)
    try:
        torch.save(ref, path)                                                                            # evaluate.py:75
    finally:
        for k, v in saved.items():
            if v is None:
                sys.modules.pop(k, None)
            else:
                sys.modules[k] = v
    assert "models.proto_refiner" not in sys.modules                     # the read below cannot lean on the reference's code
    protos = load_refiner_cache(path)
    assert isinstance(protos, list) and len(protos) == C
    assert sum(p is None for p in protos) == int((np.diff(bank.cell_off) == 0).sum()) > 0
    _arrays_equal(bank_from_protos(protos, ds_dir), bank)
    with pytest.raises(FileNotFoundError):                               # what the reference's `except FileNotFoundError` relies on
        load_refiner_cache(os.path.join(str(tmp_path), "nope.refiner"))
    torch.save({"not": "a refiner"}, path + ".bad")
    with pytest.raises(ValueError):
        load_refiner_cache(path + ".bad")


# ------------------------------------------------------------------------------------------------ metrics
def test_compute_geoguessr_metrics_matches_reference_fixture(golden_dir):
    """tests/golden/geo.npz holds the outputs of the reference's own percentage_within_radius / geoguessr_score /
    topk_geocell_accuracy / haversine_np (evaluation/metrics.py, preprocessing/geo_utils.py) on 500 seeded predictions."""
    from pigeon_amd.evaluate import compute_geoguessr_metrics
    from pigeon_amd.geo_utils import haversine_np
    g = np.load(os.path.join(golden_dir, "geo.npz"))
    preds, labels = g["metric_preds"], g["metric_labels"]
    assert np.array_equal(haversine_np(preds, labels), g["metric_distances"])
    res = compute_geoguessr_metrics((preds, g["metric_cell_preds"], None, None, None, g["metric_top5"], labels,
                                     g["metric_cell_labels"], None, None, None))
    want = dict(zip([str(k) for k in g["metric_names"]], g["metric_values"]))
    for k, v in want.items():
        assert res[k] == v, (k, res[k], v)
    assert res["Geocell_accuracy"] == float(np.mean(g["metric_cell_preds"] == g["metric_cell_labels"]))
    # one-hot labels are accepted like the reference does (metrics.py:152-158)
    oh = np.zeros((len(preds), 50)); oh[np.arange(len(preds)), g["metric_cell_labels"]] = 1
    res2 = compute_geoguessr_metrics((preds, g["metric_cell_preds"], None, None, None, g["metric_top5"], labels, oh,
                                      None, None, None))
    assert res2["Geocell_top5_accuracy"] == res["Geocell_top5_accuracy"]


@needs_reference
def test_metrics_against_live_reference_functions():
    from pigeon_amd import evaluate as ev
    mt = reference_loader.load_metrics()
    rng = np.random.default_rng(5)
    d = np.exp(rng.uniform(-3, 9, 1000))
    assert ev.geoguessr_score(d) == mt.geoguessr_score(d)
    for km in (1, 25, 750):
        assert ev.percentage_within_radius(d, km) == mt.percentage_within_radius(d, km)
    lab, top = rng.integers(0, 9, 200), rng.integers(0, 9, (200, 5))
    assert ev.topk_geocell_accuracy(lab, top) == mt.topk_geocell_accuracy(lab, top)


# ------------------------------------------------------------------------------------------------ embedding files
def test_shard_batches_pads_ragged_last_batch_like_accelerate():
    """accelerate BatchSamplerShard(split_batches=False, even_batches=True): 10 samples, batch 3, 2 ranks."""
    from pigeon_amd.distributed import shard_batches
    bat = [(torch.arange(i, min(i + 3, 10)).float()[:, None], torch.arange(i, min(i + 3, 10))) for i in range(0, 10, 3)]
    r0 = [b[1].tolist() for b in shard_batches(bat, 0, 2)]
    r1 = [b[1].tolist() for b in shard_batches(bat, 1, 2)]
    assert r0 == [[0, 1, 2], [6, 7, 8]] and r1 == [[3, 4, 5], [9, 0, 1]]
    # 11 samples, 3 ranks: short batch completed, then two filler batches cut from the wrap-around stream
    bat = [{"x": torch.arange(i, min(i + 3, 11)).float(), "index": torch.arange(i, min(i + 3, 11))} for i in range(0, 11, 3)]
    got = [[b["index"].tolist() for b in shard_batches(bat, r, 3)] for r in range(3)]
    assert got == [[[0, 1, 2], [9, 10, 0]], [[3, 4, 5], [1, 2, 3]], [[6, 7, 8], [4, 5, 6]]]
    assert all(b["x"].shape[0] == 3 for r in range(3) for b in shard_batches(bat, r, 3))


def test_compute_embeddings_writes_plain_numeric_arrays(tmp_path):
    """np.load WITHOUT allow_pickle, as the reference's reader does (dataset_preprocessing.py:294-300), incl. a ragged
    final batch in the single-process case."""
    from pigeon_amd.distributed import Communicator
    from pigeon_amd.embed import compute_embeddings
    n, bs = 11, 4
    data = [(torch.arange(i, min(i + bs, n), dtype=torch.float32)[:, None].repeat(1, 1024), torch.arange(i, min(i + bs, n)))
            for i in range(0, n, bs)]
    compute_embeddings("val", lambda px: px + 0.5, data, Communicator(), out_dir=str(tmp_path))
    embeds = np.load(os.path.join(str(tmp_path), "val.npy"))                   # no allow_pickle
    indices = np.load(os.path.join(str(tmp_path), "val_indices.npy"))
    assert embeds.dtype == np.float32 and embeds.shape == (3, 4, 1024) and indices.dtype == np.int64
    arg = np.argsort(indices.flatten())[:n]                                    # the reference reader's reorder
    e = embeds.reshape((-1, 1024))[arg]
    assert np.array_equal(e[:, 0], np.arange(n, dtype=np.float32) + 0.5)


def test_evaluate_raises_on_missing_checkpoint(tmp_path):
    from pigeon_amd.evaluate import evaluate
    with pytest.raises(FileNotFoundError):
        evaluate(os.path.join(str(tmp_path), "missing.model"), [], yfcc=False, landmarks=False, refine=False,
                 geocell_path=_geo_csv(tmp_path))


# ------------------------------------------------------------------------------------------------ fixtures are current
@needs_reference
def test_fast_golden_fixtures_regenerate_bit_identically(tmp_path, golden_dir, monkeypatch):
    """oracle/make_golden.py --only head|geo|refine re-run against /root/reference reproduces the committed files: integer /
    index arrays exactly, floating-point arrays bit for bit on the authoring CPU and to the last ulps elsewhere (the 24-layer
    fixtures take minutes and are checked by hand when they change)."""
    import importlib
    import sys
    from oracle import make_golden
    importlib.reload(make_golden)
    out = os.path.join(str(tmp_path), "golden")
    os.makedirs(out)
    monkeypatch.setattr(make_golden, "GOLD", out)
    for name in ("head", "geo", "refine"):
        monkeypatch.setattr(sys, "argv", ["make_golden.py", "--only", name])
        make_golden.main()
        a, b = np.load(os.path.join(out, f"{name}.npz")), np.load(os.path.join(golden_dir, f"{name}.npz"))
        assert sorted(a.files) == sorted(b.files), name
        for k in a.files:
            if k in ("matrix_f32x", "pairs_f32y"):          # torch's CPU fp32 cos: last ulp depends on the thread partition
                m = ~(np.isnan(a[k]) | np.isnan(b[k]))
                np.testing.assert_allclose(a[k][m], b[k][m], rtol=3e-5, atol=0.5)
                continue
            if a[k].dtype.kind == "f":
                # bit-identical on the CPU type the fixtures were written on; torch's CPU kernels choose vector width and
                # summation order by CPU, so elsewhere the last ulps of a K = 1024 fp32 dot product / an fp64 sin-cos chain move
                assert a[k].shape == b[k].shape and np.array_equal(np.isnan(a[k]), np.isnan(b[k])), (name, k)
                m = ~np.isnan(a[k])
                tol = dict(rtol=2e-5, atol=2e-4) if a[k].dtype == np.float32 else dict(rtol=1e-11, atol=1e-6)
                np.testing.assert_allclose(a[k][m], b[k][m], err_msg=f"{name}/{k}", **tol)
                continue
            assert np.array_equal(a[k], b[k]), (name, k)


def test_load_refiner_cache_committed_reference_pickle(golden_dir, tmp_path):
    """tests/golden/proto.refiner was written by the REFERENCE's ProtoRefiner through torch.save (oracle/make_golden.py --only
    refiner_cache); this runs everywhere (no /root/reference needed): the cache is readable without the reference package and
    converts to exactly the bank it was built from."""
    from pigeon_amd.proto_refiner import bank_from_protos, load_refiner_cache
    meta = np.load(os.path.join(golden_dir, "refiner_cache.npz"))
    C, ppc, bseed, maxm = [int(x) for x in meta["meta"]]
    assert "models.proto_refiner" not in sys.modules
    protos = load_refiner_cache(os.path.join(golden_dir, "proto.refiner"))
    assert len(protos) == C and sum(p is None for p in protos) == int(meta["n_empty"])
    bank = synthetic.make_bank(C, ppc, seed=bseed, empty_frac=0.05, max_members=maxm)
    ds_dir = os.path.join(str(tmp_path), "hf_train")
    synthetic.write_bank_reference_files(bank, os.path.join(str(tmp_path), "protos.csv"), ds_dir)
    _arrays_equal(bank_from_protos(protos, ds_dir), bank)


class _CacheHolder(torch.nn.Module):
    """pickled under the reference's class name (see _save_as_reference_cache)"""


_CacheHolder.__module__, _CacheHolder.__qualname__, _CacheHolder.__name__ = "models.proto_refiner", "ProtoRefiner", "ProtoRefiner"


class _Evil:
    marker = None

    def __reduce__(self):
        return (os.system, (f"touch {_Evil.marker}",))


class _ViaLoadFromBytes:
    """unpickles as torch.storage._load_from_bytes(payload)"""

    def __init__(self, payload):
        self.payload = payload

    def __reduce__(self):
        return (torch.storage._load_from_bytes, (self.payload,))


def _save_as_reference_cache(protos, path, protocol=2):
    import types
    h = _CacheHolder()
    h.protos = protos
    mods = {"models": types.ModuleType("models"), "models.proto_refiner": types.ModuleType("models.proto_refiner")}
    mods["models.proto_refiner"].ProtoRefiner = _CacheHolder
    saved = {k: sys.modules.get(k) for k in mods}
    sys.modules.update(mods)
    try:
        torch.save(h, path, pickle_protocol=protocol)
    finally:
        for k, v in saved.items():
            if v is None:
                sys.modules.pop(k, None)
            else:
                sys.modules[k] = v
    return path


def test_load_refiner_cache_refuses_what_a_cache_does_not_hold(tmp_path):
    """A pickle executes what it names: the loader's allow-list must refuse a global outside a refiner cache's data model -- directly,
    and NESTED inside `torch.storage._load_from_bytes` (torch's own implementation of that function unpickles its argument with the
    stock unpickler, which would resolve anything: ADVICE r05) -- while a legitimate `_load_from_bytes` payload (a torch-saved tensor)
    and a protocol-0 object graph (`copyreg._reconstructor(cls, object, None)`) still load."""
    import io
    import pickle
    from pigeon_amd.proto_refiner import load_refiner_cache
    _Evil.marker = os.path.join(str(tmp_path), "executed")
    d = str(tmp_path)

    def expect_refused(path):
        with pytest.raises(pickle.UnpicklingError):
            load_refiner_cache(path)
        assert not os.path.exists(_Evil.marker)

    expect_refused(_save_as_reference_cache([_Evil()], os.path.join(d, "direct.refiner")))
    expect_refused(_save_as_reference_cache([_ViaLoadFromBytes(pickle.dumps(_Evil()))], os.path.join(d, "nested.refiner")))
    buf = io.BytesIO()
    torch.save([_Evil()], buf)                                     # the nested payload as a torch zip archive: same refusal
    expect_refused(_save_as_reference_cache([_ViaLoadFromBytes(buf.getvalue())], os.path.join(d, "nested_zip.refiner")))
    # what a cache may legitimately hold still loads: a tensor that travels through _load_from_bytes ...
    t = torch.arange(6, dtype=torch.float32).reshape(2, 3)
    buf = io.BytesIO()
    torch.save(t, buf)
    got = load_refiner_cache(_save_as_reference_cache([_ViaLoadFromBytes(buf.getvalue()), None], os.path.join(d, "tensor.refiner")))
    assert torch.equal(got[0], t) and got[1] is None
    # ... and a protocol-0 pickle (copyreg._reconstructor + builtins.object for the module shell)
    got = load_refiner_cache(_save_as_reference_cache([None, {"a": (1, 2.5, "x")}], os.path.join(d, "proto0.refiner"), protocol=0))
    assert got == [None, {"a": (1, 2.5, "x")}]


def test_pretrained_tower_from_local_files(tmp_path, monkeypatch):
    """reference models/clip_embedder.py:25-26 and evaluation/evaluate.py:36-40 call `CLIPVisionModel.from_pretrained(CLIP_MODEL)`;
    offline the same call runs with local_files_only=True against a `save_pretrained` directory (env PIGEON_CLIP_MODEL) or the HF
    cache.  A one-layer tower of the ViT-L/14-336 geometry stands in for the hub checkpoint."""
    from transformers import CLIPVisionConfig, CLIPVisionModel
    from pigeon_amd import clip_embedder, synthetic
    from pigeon_amd.clip_embedder import CLIPEmbedding, HipCLIPVisionModel, load_pretrained_clip
    cfg = CLIPVisionConfig(hidden_size=1024, intermediate_size=4096, num_hidden_layers=1, num_attention_heads=16, image_size=336,
                           patch_size=14, projection_dim=768)
    hf = CLIPVisionModel(cfg)
    sd = synthetic.make_vit_weights(seed=3, layers=1, affine_jitter=True)
    hf.load_state_dict(sd, strict=True)
    d = str(tmp_path / "clip_local")
    hf.save_pretrained(d)
    m = load_pretrained_clip(d)
    assert isinstance(m, HipCLIPVisionModel) and m.config.num_hidden_layers == 1
    got = m.state_dict()
    for k, v in sd.items():
        assert torch.equal(got[k], v), k
    # nothing cached under the hub id, no directory: the reason is in the message
    monkeypatch.delenv("PIGEON_CLIP_MODEL", raising=False)
    with pytest.raises(RuntimeError, match="local_files_only"):
        load_pretrained_clip()
    with pytest.raises(RuntimeError, match="PIGEON_CLIP_MODEL"):
        CLIPEmbedding("openai/clip-vit-large-patch14-336", device="cpu")
    # the reference's constructor call, resolved through the environment
    monkeypatch.setenv("PIGEON_CLIP_MODEL", d)
    emb = CLIPEmbedding("openai/clip-vit-large-patch14-336", device="cpu")
    assert torch.equal(emb.clip_model.state_dict()["encoder.layers.0.mlp.fc1.bias"], sd["encoder.layers.0.mlp.fc1.bias"])
    # ... and with a checkpoint copied over it the reference's way (embedder=True strips `base_model.`)
    ck = {"base_model." + k: v + 1 for k, v in sd.items() if k.startswith("encoder.layers.0.mlp")}
    ckp = str(tmp_path / "embedder.ckpt")
    torch.save(ck, ckp)
    emb2 = CLIPEmbedding(ckp, device="cpu", load_checkpoint=True)
    assert torch.equal(emb2.clip_model.state_dict()["encoder.layers.0.mlp.fc1.bias"], sd["encoder.layers.0.mlp.fc1.bias"] + 1)
    assert torch.equal(emb2.clip_model.state_dict()["pre_layrnorm.weight"], sd["pre_layrnorm.weight"])
    # a tower of another geometry is refused by name
    small = CLIPVisionModel(CLIPVisionConfig(hidden_size=64, intermediate_size=128, num_hidden_layers=1, num_attention_heads=2,
                                             image_size=28, patch_size=14, projection_dim=32))
    d2 = str(tmp_path / "clip_small")
    small.save_pretrained(d2)
    with pytest.raises(RuntimeError, match="ViT-L/14-336"):
        load_pretrained_clip(d2)


def test_evaluate_accepts_the_references_base_model_strings(tmp_path, monkeypatch):
    """reference evaluation/evaluate.py:10-12,36-40: `base_model` is a STRING -- CLIP_MODEL or a checkpoint path.  The string branch
    is resolved before any GPU work (the SuperGuessr it builds then needs the GPU: stopped there on this box)."""
    from transformers import CLIPVisionConfig, CLIPVisionModel
    from pigeon_amd import config as cfg, evaluate as ev, synthetic
    sd = synthetic.make_vit_weights(seed=4, layers=1)
    hf = CLIPVisionModel(CLIPVisionConfig(hidden_size=1024, intermediate_size=4096, num_hidden_layers=1, num_attention_heads=16,
                                          image_size=336, patch_size=14, projection_dim=768))
    hf.load_state_dict(sd, strict=True)
    d = str(tmp_path / "clip_local")
    hf.save_pretrained(d)
    seen = {}

    class Stop(Exception):
        pass

    def fake_sg(base_model, **kw):
        seen["base"] = base_model
        raise Stop()
    monkeypatch.setattr(ev, "SuperGuessr", fake_sg)
    monkeypatch.setenv("PIGEON_CLIP_MODEL", d)
    with pytest.raises(Stop):
        ev.evaluate("none", None, False, False, base_model=cfg.CLIP_MODEL)
    assert torch.equal(seen["base"].state_dict()["pre_layrnorm.bias"], sd["pre_layrnorm.bias"])
    ck = str(tmp_path / "base.ckpt")
    torch.save({"vision_model.pre_layrnorm.bias": sd["pre_layrnorm.bias"] + 2}, ck)
    with pytest.raises(Stop):
        ev.evaluate("none", None, False, False, base_model=ck)
    assert torch.equal(seen["base"].state_dict()["pre_layrnorm.bias"], sd["pre_layrnorm.bias"] + 2)
    # no pretrained tower on the machine: a checkpoint that carries the whole tower still works, a partial one is refused
    monkeypatch.delenv("PIGEON_CLIP_MODEL")
    full = str(tmp_path / "full.ckpt")
    torch.save({"base_model." + k: v for k, v in sd.items()}, full)
    with pytest.raises(Stop):
        ev.evaluate("none", None, False, False, base_model=full)
    assert torch.equal(seen["base"].state_dict()["encoder.layers.0.mlp.fc2.bias"], sd["encoder.layers.0.mlp.fc2.bias"])
    with pytest.raises(RuntimeError, match="complete vision tower"):
        ev.evaluate("none", None, False, False, base_model=ck)
    with pytest.raises(RuntimeError, match="local_files_only"):
        ev.evaluate("none", None, False, False, base_model=cfg.CLIP_MODEL)


def test_gemm_routing_model_against_the_measured_sweep():
    """pg_gemm_route (host arithmetic, no launch): which of the three bit-identical GEMM kernels a launch of up to ~64 images takes.
    Checked against the sweeps the model was fitted to (profiles/r06/gemm_three_sweep.txt, gemm_three_sweep_producer.txt: the model's four GEMM shapes x 1 .. 64 images
    x 384 x 256 persistent / 256 x 256 persistent / gemm_mid.hip, microseconds on MI355X): the pick is never more than 10 % off the
    measured best of its cell (the 256 x 256 kernel must win by 10 % in the model to be taken -- in a forward the ties went the other
    way, profiles/r06/latency_route_ab.txt); the benchmark's 512-image launches keep the variant's own kernel; switching the routing off
    does what it says."""
    import ctypes as C
    import re
    from pigeon_amd import _lib
    L = _lib.load()

    def route(variant, epi, M, N, K):
        k = C.c_int(-9)
        assert L.pg_gemm_route(variant, epi, M, N, K, C.byref(k)) == 0
        return k.value

    shapes = {"qkv": (3072, 1024, _lib.EPI_QKV_LN), "out": (1024, 1024, _lib.EPI_RESID_STAT), "fc1": (4096, 1024, _lib.EPI_GELU_LN),
              "fc2": (1024, 4096, _lib.EPI_RESID_STAT)}
    ns = (1, 2, 4, 8, 12, 16, 24, 32, 64)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

    def against(sweep, only=None):
        picks, worst, seen = {}, 1.0, 0
        for line in open(os.path.join(root, "profiles", "r06", sweep)):
            name = line.split()[0] if line.strip() else ""
            if name not in shapes or (only and name not in only):
                continue
            N, K, epi = shapes[name]
            cells = re.findall(r"([\d.]+)/\s*([\d.]+)/\s*([\d.]+)[pqm]", line)
            assert len(cells) == len(ns), (name, len(cells))
            for n, cell in zip(ns, cells):
                t = [float(v) for v in cell]                   # [variant 56's own kernel, 256 x 256, gemm_mid]
                kind = route(56, epi, 577 * n, N, K)
                picks[(name, n)] = kind
                took = t[2] if kind == 2 else (t[0] if (kind == 0 or name == "out") else t[1])     # (out-projection: own kernel IS 256 x 256)
                worst = max(worst, took / min(t))
                seen += 1
        return picks, worst, seen
    # the first session's sweep: its gemm_mid column is the kernel BEFORE the producer wave, i.e. an upper bound for the cells that take it
    picks, worst, seen = against("gemm_three_sweep.txt")
    assert seen == 36 and worst < 1.10, worst
    # the sweep with the producer wave: the two residual shapes (its LN-fold rows carry the standalone loop's artefact, profiles/r06/README.md)
    _, worst2, seen2 = against("gemm_three_sweep_producer.txt", only=("out", "fc2"))
    assert seen2 == 18 and worst2 < 1.05, worst2
    assert picks[("qkv", 1)] == 2 and picks[("qkv", 16)] == 1 and picks[("qkv", 12)] == 0 and picks[("fc1", 4)] == 1 and picks[("qkv", 4)] == 2
    assert picks[("fc2", 4)] == 2 and picks[("fc2", 16)] == 1 and picks[("fc2", 32)] == 0 and picks[("out", 4)] == 2 and picks[("out", 16)] == 1
    assert picks[("out", 8)] == 2 and picks[("out", 12)] == 2          # (the 256 x 256 kernel until the producer wave made gemm_mid the faster one there)
    # the benchmark step (295 424 rows) is outside the routed range: the variant's own kernel (fc2 / QKV / fc1 384 x 256, out-projection 256 x 256)
    assert [route(56, shapes[s][2], 295424, shapes[s][0], shapes[s][1]) for s in ("qkv", "out", "fc1", "fc2")] == [0, 1, 0, 0]
    assert route(36, _lib.EPI_F32, 16156, 1024, 3072) == 1 and route(8, _lib.EPI_QKV, 577, 3072, 1024) == -1
    try:
        assert L.pg_tune_gemm_mid(0) == 0
        assert route(56, _lib.EPI_QKV_LN, 577, 3072, 1024) == 0 and route(56, _lib.EPI_RESID_STAT, 577, 1024, 1024) == 1
        assert L.pg_tune_gemm_mid(2) == 0                       # gemm_mid.hip the only alternative
        assert route(56, _lib.EPI_QKV_LN, 16 * 577, 3072, 1024) == 0 and route(56, _lib.EPI_QKV_LN, 577, 3072, 1024) == 2
    finally:
        L.pg_tune_gemm_mid(1)
