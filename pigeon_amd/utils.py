"""ModelOutput and name-wise state-dict loading, as reference models/utils.py."""
from collections import namedtuple
from typing import Dict

from torch.nn.parameter import Parameter

# reference models/utils.py:7-9 (12 fields, same order)
ModelOutput = namedtuple('ModelOutput', 'loss loss_clf loss_reg loss_climate loss_month '
                         'preds_LLH preds_geocell preds_mt preds_climate preds_month '
                         'top5_geocells embedding')

TopK = namedtuple('topk', 'values indices')   # what torch.topk returns (models/super_guessr.py:459)


def resolve_name(name: str, own_state) -> str:
    """Map a checkpoint parameter name onto this package's names.  Reference checkpoints were written with
    transformers 4.23.1 (env.yml:60), whose CLIPVisionModel nests everything under `vision_model.`
    (`base_model.vision_model.encoder.layers...` inside a SuperGuessr checkpoint); HipCLIPVisionModel exposes the flat
    transformers >= 5 names.  The `vision_model.` component is dropped when that makes the name resolvable."""
    if name in own_state:
        return name
    parts = name.split('.')
    if 'vision_model' in parts:
        flat = '.'.join(p for p in parts if p != 'vision_model')
        if flat in own_state:
            return flat
    return name


def load_state_dict(self, state_dict: Dict, embedder: bool = False):
    """Loads parameters in state_dict into model wherever possible (reference models/utils.py:24-45):
    copies by name, skipping unknown names with a message; with embedder=True a leading dotted component
    containing 'base_model' is stripped (:34-35).  Unlike the reference, a checkpoint of which NOT A SINGLE name
    matches raises: silently keeping random weights is never what the caller wanted."""
    own_state = self.state_dict()
    matched = 0
    for name, param in state_dict.items():
        if embedder and 'base_model' in name:
            name = '.'.join(name.split('.')[1:])
        name = resolve_name(name, own_state)
        if name not in own_state:
            print(f'Parameter {name} not in model\'s state.')
            continue
        if isinstance(param, Parameter):
            param = param.data
        own_state[name].copy_(param)
        matched += 1
    if len(state_dict) > 0 and matched == 0:
        raise KeyError(f'load_state_dict: none of the {len(state_dict)} checkpoint parameters matched the model '
                       f'(first key: {next(iter(state_dict))!r})')
    if hasattr(self, '_weights_changed'):
        self._weights_changed()
    return matched
