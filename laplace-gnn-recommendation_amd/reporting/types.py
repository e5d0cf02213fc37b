"""Result records returned by the pipelines (reference: reporting/types.py:29-35)."""
from dataclasses import dataclass


@dataclass
class Stats:
    loss: float
    recall_val: float
    recall_test: float
    precision_val: float
    precision_test: float
