"""Import the REAL PIGEON hot-path modules from /root/reference with environment shims.

TEST INFRASTRUCTURE ONLY. Nothing under ``pigeon_amd/`` may import this module; it is used by
``oracle/make_golden.py`` (fixture generation, authoring container only) and by the CPU tests that
pin ``oracle/pigeon_oracle.py`` (the restatement) against the reference itself when
``/root/reference`` exists.  On the GPU box the reference tree is absent and ``available()`` is False.

Why shims are needed (all are environment drift, none change arithmetic):
  * ``config.py:1,94-177`` builds ``transformers.TrainingArguments(evaluation_strategy=...)`` which raises
    on transformers 5.x -> we exec only ``config.py:1-92`` (the constants) without the transformers import.
  * ``preprocessing/__init__.py:2-3`` pulls in geopandas/srtm/... -> we expose only ``geo_utils.py`` and
    ``utils.py`` (both import just numpy/torch/PIL/config).
  * ``models/proto_refiner.py`` hard-codes ``'cuda'`` (:172,176,187,198,225,229,230) and relies on
    datasets==2.6.1 returning a tensor for ``dataset['embedding']`` (:176) where datasets>=4 returns a lazy
    ``Column`` -> two textual substitutions applied to the source before exec, when device == 'cpu'.
  * ``ProtoRefiner.load_prototypes`` (:257-286) forks 64 processes -> replaced by a serial loop over the
    reference's own ``_get_prototypes`` (:288-313).
"""
import importlib.util
import os
import sys
import types

REFERENCE_ROOT = os.environ.get("PIGEON_REFERENCE_ROOT", "/root/reference")


def available() -> bool:
    return os.path.isfile(os.path.join(REFERENCE_ROOT, "models", "proto_refiner.py"))


def _load_file_as(name: str, path: str, patch=None):
    with open(path, "r") as f:
        src = f.read()
    if patch is not None:
        src = patch(src)
    mod = types.ModuleType(name)
    mod.__file__ = path
    sys.modules[name] = mod
    exec(compile(src, path, "exec"), mod.__dict__)
    return mod


def load(geocell_path: str, proto_path: str, dataset_path: str, device: str = "cpu"):
    """Returns a namespace with the reference's own classes/functions.

    geocell_path / proto_path / dataset_path override ``config.GEOCELL_PATH`` (config.py:35),
    ``PROTO_PATH`` (:76) and ``DATASET_PATH`` (:77) so the reference reads our synthetic fixtures.
    """
    if not available():
        raise RuntimeError(f"reference tree not found at {REFERENCE_ROOT}")

    saved = {k: sys.modules.get(k) for k in
             ("config", "preprocessing", "models", "models.layers", "models.utils",
              "models.super_guessr", "models.proto_refiner", "models.clip_embedder",
              "models.layers.hedge", "models.layers.positional_encoder")}

    # --- config: constants only (config.py:1-92) -------------------------------------------------
    with open(os.path.join(REFERENCE_ROOT, "config.py")) as f:
        cfg_src = f.read()
    cfg_src = cfg_src.split("# Training arguments")[0]
    cfg_src = cfg_src.replace("from transformers import TrainingArguments", "")
    config = types.ModuleType("config")
    exec(compile(cfg_src, "config.py", "exec"), config.__dict__)
    config.GEOCELL_PATH = geocell_path
    config.GEOCELL_PATH_YFCC = geocell_path
    config.PROTO_PATH = proto_path
    config.DATASET_PATH = dataset_path
    sys.modules["config"] = config

    # --- preprocessing: geo_utils + utils only ---------------------------------------------------
    prep = types.ModuleType("preprocessing")
    prep.__path__ = []
    sys.modules["preprocessing"] = prep
    for fname in ("geo_utils.py", "utils.py"):
        m = _load_file_as("preprocessing." + fname[:-3], os.path.join(REFERENCE_ROOT, "preprocessing", fname))
        for k, v in m.__dict__.items():
            if not k.startswith("_"):
                setattr(prep, k, v)

    # --- models package, file by file ------------------------------------------------------------
    models = types.ModuleType("models")
    models.__path__ = [os.path.join(REFERENCE_ROOT, "models")]
    sys.modules["models"] = models
    layers = types.ModuleType("models.layers")
    layers.__path__ = [os.path.join(REFERENCE_ROOT, "models", "layers")]
    sys.modules["models.layers"] = layers
    hedge = _load_file_as("models.layers.hedge", os.path.join(REFERENCE_ROOT, "models", "layers", "hedge.py"))
    posenc = _load_file_as("models.layers.positional_encoder",
                           os.path.join(REFERENCE_ROOT, "models", "layers", "positional_encoder.py"))
    layers.HedgeLayer = hedge.HedgeLayer
    layers.PositionalEncoder = posenc.PositionalEncoder

    utils = _load_file_as("models.utils", os.path.join(REFERENCE_ROOT, "models", "utils.py"))
    sg = _load_file_as("models.super_guessr", os.path.join(REFERENCE_ROOT, "models", "super_guessr.py"))

    def patch_refiner(src: str) -> str:
        if device == "cpu":
            src = src.replace("'cuda'", "'cpu'")
        # datasets>=4 lazy Column -> materialise (datasets 2.6.1 returned the tensor directly)
        src = src.replace("cell_emb['embedding'].to(", "cell_emb['embedding'][:].to(")
        return src

    pr = _load_file_as("models.proto_refiner", os.path.join(REFERENCE_ROOT, "models", "proto_refiner.py"),
                       patch=patch_refiner)

    def patch_embedder(src: str) -> str:
        return src.replace("from .utils import load_state_dict", "from models.utils import load_state_dict")

    ce = _load_file_as("models.clip_embedder", os.path.join(REFERENCE_ROOT, "models", "clip_embedder.py"),
                       patch=patch_embedder)

    def serial_load_prototypes(self):
        # serial stand-in for proto_refiner.py:257-286 using the reference's own _get_prototypes
        import datasets as _ds
        _ds.disable_progress_bar()
        self.protos = [self._get_prototypes(i) for i in range(self.num_geocells)]
        _ds.enable_progress_bar()

    pr.ProtoRefiner.load_prototypes = serial_load_prototypes

    ns = types.SimpleNamespace(
        config=config, preprocessing=prep,
        SuperGuessr=sg.SuperGuessr, ProtoRefiner=pr.ProtoRefiner, CLIPEmbedding=ce.CLIPEmbedding,
        ModelOutput=utils.ModelOutput, load_state_dict=utils.load_state_dict,
        haversine=prep.haversine, haversine_np=prep.haversine_np, haversine_matrix=prep.haversine_matrix,
    )

    # leave sys.modules clean so pigeon_amd's own modules are never shadowed
    for k, v in saved.items():
        if v is None:
            sys.modules.pop(k, None)
        else:
            sys.modules[k] = v
    for k in list(sys.modules):
        if k.startswith("preprocessing.") or k.startswith("models."):
            sys.modules.pop(k, None)
    return ns


def load_metrics():
    """The reference's metric helpers (evaluation/metrics.py:89-136: percentage_within_radius, geoguessr_score,
    topk_geocell_accuracy) plus haversine_np, WITHOUT importing the module: its top level pulls in geopandas / sklearn /
    config.TRAIN_ARGS.  The three functions are cut out of the source by name (ast) and exec'd next to numpy and
    DECAY_CONSTANT (config.py:52) -- their bodies run unmodified."""
    import ast
    import numpy as np
    if not available():
        raise RuntimeError(f"reference tree not found at {REFERENCE_ROOT}")
    path = os.path.join(REFERENCE_ROOT, "evaluation", "metrics.py")
    src = open(path).read()
    tree = ast.parse(src)
    want = ("percentage_within_radius", "geoguessr_score", "topk_geocell_accuracy")
    with open(os.path.join(REFERENCE_ROOT, "config.py")) as f:
        cfg = {}
        exec(compile(f.read().split("# Training arguments")[0].replace("from transformers import TrainingArguments", ""),
                     "config.py", "exec"), cfg)
    env = {"np": np, "DECAY_CONSTANT": cfg["DECAY_CONSTANT"], "Dict": dict}
    for node in tree.body:
        if isinstance(node, ast.FunctionDef) and node.name in want:
            exec(compile(ast.Module(body=[node], type_ignores=[]), path, "exec"), env)
    geo = _load_file_as("_ref_geo_utils_tmp", os.path.join(REFERENCE_ROOT, "preprocessing", "geo_utils.py"))
    sys.modules.pop("_ref_geo_utils_tmp", None)
    return types.SimpleNamespace(haversine_np=geo.haversine_np, **{k: env[k] for k in want})


def make_reference_embedder(ns, vit, device="cpu", panorama=False):
    """Instantiate the reference CLIPEmbedding around an existing HF CLIPVisionModel.

    ``CLIPEmbedding.__init__`` (models/clip_embedder.py:11-40) calls ``from_pretrained`` on the hub, which is
    impossible offline; we build the object without it and then run the reference's own ``forward``
    (:79-89) / ``_get_embedding`` (:42-66) with tensor input.
    """
    import torch
    emb = ns.CLIPEmbedding.__new__(ns.CLIPEmbedding)
    torch.nn.Module.__init__(emb)
    emb.device = device
    emb.processor = None
    emb.clip_model = vit
    emb.panorama = panorama
    emb.eval()
    return emb
