"""Feature bookkeeping for the ranker (reference: utils/get_info.py:10-48)."""
from typing import Dict, Tuple

import torch as t
from torch import Tensor

from ..config import embedding_range_dict
from ..data.types import FeatureInfo
from ..utils.constants import Constants


def embedding_size_for(max_category: int) -> int:
    """First bucket whose bound covers the cardinality; beyond the last bucket the reference falls
    back to the "10000" width (20) — kept as is."""
    for bound, width in embedding_range_dict.items():
        if max_category <= int(bound):
            return width
    return embedding_range_dict["10000"]


def get_feature_info(full_data) -> Dict[str, FeatureInfo]:
    info = {}
    node_types, _ = full_data.metadata()
    for node_type in node_types:
        x = full_data.x_dict[node_type]
        num_cat = t.max(x, dim=0)[0].tolist()
        info[node_type] = FeatureInfo(num_feat=x.shape[1], num_cat=num_cat,
                                      embedding_size=[embedding_size_for(m) for m in num_cat])
    return info


def select_properties(data) -> Tuple[dict, dict, Tensor, Tensor]:
    store = data[Constants.edge_key]
    return data.x_dict, data.edge_index_dict, store.edge_label_index, store.edge_label.float()
