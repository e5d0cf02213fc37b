"""Feature bookkeeping for the ranker: which embedding width each categorical column gets, and the four things a
training / evaluation step reads from a batch.  Same results as the reference's utils/get_info.py:10-48 (checked
against tests/golden/config_defaults.pt and the ranker parity tests); the widths are found for all columns of a node
type at once with a sorted-bounds lookup instead of a per-column scan of the table."""
from bisect import bisect_left
from typing import Dict, List, Tuple

from torch import Tensor

from ..config import embedding_range_dict
from ..data.types import FeatureInfo
from ..utils.constants import Constants

# (upper bound of the cardinality, embedding width), ascending — config.py's table with integer keys
_BOUNDS: List[int] = sorted(int(k) for k in embedding_range_dict)
_WIDTH_OF_BOUND = {int(k): v for k, v in embedding_range_dict.items()}
# A cardinality above the last bound gets the width listed under "10000" — 20, not the largest width: the
# reference's fall-through as written.
_OVERFLOW_WIDTH = embedding_range_dict["10000"]


def embedding_size_for(max_category: int) -> int:
    slot = bisect_left(_BOUNDS, int(max_category))   # first bound >= max_category
    return _WIDTH_OF_BOUND[_BOUNDS[slot]] if slot < len(_BOUNDS) else _OVERFLOW_WIDTH


def get_feature_info(full_data) -> Dict[str, FeatureInfo]:
    """Per node type: number of categorical columns, the largest code of each, the width chosen for each."""
    described: Dict[str, FeatureInfo] = {}
    for node_type in full_data.metadata()[0]:
        codes = full_data.x_dict[node_type]
        largest = codes.amax(dim=0).tolist()
        described[node_type] = FeatureInfo(num_feat=codes.shape[1], num_cat=largest,
                                           embedding_size=[embedding_size_for(c) for c in largest])
    return described


def select_properties(data) -> Tuple[dict, dict, Tensor, Tensor]:
    """(x_dict, edge_index_dict, label pairs of the `buys` relation, their labels as float)."""
    labelled = data[Constants.edge_key]
    return data.x_dict, data.edge_index_dict, labelled.edge_label_index, labelled.edge_label.float()
