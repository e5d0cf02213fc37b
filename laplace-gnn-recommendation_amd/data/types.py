"""Record types shared by the data layer (reference: data/types.py:56-63)."""
from dataclasses import dataclass
from typing import List

ArticleIdMap = dict
CustomerIdMap = dict


@dataclass
class FeatureInfo:
    num_feat: int
    num_cat: List[int]
    embedding_size: List[int]
