from typing import Any, Dict, Optional

import torch


def set_weight_attrs(weight: torch.Tensor, weight_attrs: Optional[Dict[str, Any]]) -> None:
    """Attach loader metadata to a parameter; an attribute that already exists is never replaced
    (vllm/model_executor/utils.py:13-43)."""
    if not weight_attrs:
        return
    for key, value in weight_attrs.items():
        assert not hasattr(weight, key), f"Overwriting existing tensor attribute: {key}"
        setattr(weight, key, value)
