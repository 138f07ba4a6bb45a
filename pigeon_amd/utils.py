"""ModelOutput and name-wise state-dict loading, as reference models/utils.py."""
from collections import namedtuple
from typing import Dict

from torch.nn.parameter import Parameter

# reference models/utils.py:7-9 (12 fields, same order)
ModelOutput = namedtuple('ModelOutput', 'loss loss_clf loss_reg loss_climate loss_month '
                         'preds_LLH preds_geocell preds_mt preds_climate preds_month '
                         'top5_geocells embedding')

TopK = namedtuple('topk', 'values indices')   # what torch.topk returns (models/super_guessr.py:459)


def load_state_dict(self, state_dict: Dict, embedder: bool = False):
    """Loads parameters in state_dict into model wherever possible (reference models/utils.py:24-45):
    copies by name, skipping unknown names with a message; with embedder=True a leading dotted component
    containing 'base_model' is stripped (:34-35)."""
    own_state = self.state_dict()
    for name, param in state_dict.items():
        if embedder and 'base_model' in name:
            name = '.'.join(name.split('.')[1:])
        if name not in own_state:
            print(f'Parameter {name} not in model\'s state.')
            continue
        if isinstance(param, Parameter):
            param = param.data
        own_state[name].copy_(param)
    if hasattr(self, '_weights_changed'):
        self._weights_changed()
