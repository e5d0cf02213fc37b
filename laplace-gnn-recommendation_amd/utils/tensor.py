"""Small tensor helpers with the reference's semantics (utils/tensor.py:16-61), checked against
tests/golden/tensor_utils.pt."""
from typing import List, Union

import numpy as np
import torch as t
import torch.nn.functional as F
from torch import Tensor


def difference_1d(a: Tensor, b: Tensor, assume_unique: bool) -> Tensor:
    """Elements of `a` not in `b`.  With assume_unique=True the order of `a` is preserved
    (numpy.setdiff1d semantics) — the property top-K exclusion relies on."""
    av, bv = a.detach().cpu().numpy(), b.detach().cpu().numpy()
    return t.tensor(np.setdiff1d(av, bv, assume_unique=assume_unique))


def padded_stack(tensors: List[Tensor], side: str = "right", mode: str = "constant",
                 value: Union[int, float] = 0) -> Tensor:
    """Stack 1-D (or [..., L]) tensors after padding the last dimension to the longest."""
    if side not in ("left", "right"):
        raise ValueError(f"side for padding '{side}' is unknown")
    width = max(x.size(-1) for x in tensors)
    rows = []
    for x in tensors:
        gap = width - x.size(-1)
        if gap > 0:
            x = F.pad(x, (gap, 0) if side == "left" else (0, gap), mode=mode, value=value)
        rows.append(x)
    return t.stack(rows, dim=0)
