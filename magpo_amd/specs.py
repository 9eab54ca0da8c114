"""Minimal value specs of the MarlEnv contract (jumanji.specs as used by mava/types.py:86-123 and the wrappers' observation_spec /
action_spec: mava/wrappers/jumanji.py:102-135, mava/coordsum/env.py:148-183, mava/wrappers/observation.py:60-80).

Only what the system file and the network builders read: shape, dtype, bounds / num_values, name, ``replace`` and
``generate_value``.  Shapes are PER ENV (no batch axis), as in the reference."""
from __future__ import annotations

from typing import Any, Callable, Dict, Tuple

import numpy as np


class Array:
    def __init__(self, shape: Tuple[int, ...], dtype, name: str = ""):
        self.shape, self.dtype, self.name = tuple(int(s) for s in shape), np.dtype(dtype), name

    def replace(self, **kw) -> "Array":
        new = self.__class__.__new__(self.__class__)
        new.__dict__.update(self.__dict__)
        for k, v in kw.items():
            setattr(new, k, np.dtype(v) if k == "dtype" else v)
        return new

    def generate_value(self) -> np.ndarray:
        return np.zeros(self.shape, self.dtype)

    def __repr__(self):
        return f"{type(self).__name__}(shape={self.shape}, dtype={self.dtype}, name={self.name!r})"


class BoundedArray(Array):
    def __init__(self, shape, dtype, minimum, maximum, name: str = ""):
        super().__init__(shape, dtype, name)
        self.minimum, self.maximum = np.asarray(minimum), np.asarray(maximum)

    def generate_value(self) -> np.ndarray:
        return np.broadcast_to(self.minimum, self.shape).astype(self.dtype)


class DiscreteArray(BoundedArray):
    def __init__(self, num_values: int, dtype=np.int32, name: str = ""):
        super().__init__((), dtype, 0, int(num_values) - 1, name)
        self.num_values = int(num_values)


class MultiDiscreteArray(BoundedArray):
    def __init__(self, num_values, dtype=np.int32, name: str = ""):
        nv = np.asarray(num_values, np.int32)
        super().__init__(nv.shape, dtype, np.zeros_like(nv), nv - 1, name)
        self.num_values = nv


class Spec:
    """A named tree of specs with a constructor (jumanji specs.Spec(Observation, "ObservationSpec", **fields))."""

    def __init__(self, constructor: Callable[..., Any], name: str = "", **specs: Any):
        self._constructor, self.name, self._specs = constructor, name, dict(specs)
        for k, v in specs.items():
            setattr(self, k, v)

    def replace(self, **kw) -> "Spec":
        return Spec(self._constructor, self.name, **{**self._specs, **kw})

    def generate_value(self):
        return self._constructor(**{k: v.generate_value() for k, v in self._specs.items()})

    @property
    def fields(self) -> Dict[str, Any]:
        return dict(self._specs)

    def __repr__(self):
        return f"Spec({self.name!r}, {self._specs})"
