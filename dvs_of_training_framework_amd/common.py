"""Small helpers with the reference's semantics (utils/common.py)."""
import numpy as np
import torch


def mean(values):
    """utils/common.py:22-23"""
    return sum(values) / len(values)


def to_tensor(data):
    """utils/common.py:240-259: numpy int arrays -> long, anything else ->
    float32; dictionaries are converted item by item."""
    if isinstance(data, dict):
        for k, v in data.items():
            data[k] = to_tensor(v)
        return data
    if isinstance(data, np.ndarray) and data.dtype == np.int_:
        return torch.tensor(data, dtype=torch.long)
    return torch.tensor(data, dtype=torch.float32)
