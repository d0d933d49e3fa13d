"""Host helpers mirrored from the reference."""
from __future__ import annotations

import math
from typing import Any

from torch import Tensor


def reshape_apply(x: Tensor, layer: Any) -> Tensor:
    """Flatten every leading dim, call the 2-D-only layer, restore the shape
    (/root/reference/language_interpolation/utils.py:21-37).  The element count
    is computed on the host, without the reference's `torch.prod(torch.tensor(...))`
    device round trip."""
    shape = x.shape
    first = int(math.prod(shape[:-1]))
    flatout: Tensor = layer(x.reshape(first, shape[-1]))
    return flatout.view(*shape[:-1], -1)
