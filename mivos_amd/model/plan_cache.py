"""Shared bookkeeping of the networks' compiled plans (packed / pre-split weights living on the GPU).

A plan is built lazily by ``plan()`` and must be dropped whenever the parameters it was packed from change:

* ``load_state_dict`` and device / dtype moves (``.to``, ``.cuda``) drop it themselves - a ``.to(device)`` of a network that
  already lives there keeps every storage and therefore the plan (InferenceCore / DAVISProcessor re-apply ``.to`` per clip);
* ``refresh_plan_if_stale()`` (called by InferenceCore and DAVISProcessor once per clip) compares a fingerprint of every
  parameter and buffer - ``(data_ptr, _version)`` - with the one taken when the plan was built: optimiser steps, ``p.copy_``,
  ``p.mul_`` ... bump ``_version``, re-assigned parameters change ``data_ptr``;
* writes through ``p.data`` (``p.data.copy_(...)``, ``p.data = ...``) bypass autograd's version counter BY DESIGN and cannot be
  seen this way: call ``invalidate_plan()`` after them.
"""
import torch.nn as nn


class PlanCache(nn.Module):
    def __init__(self):
        super().__init__()
        self._plan = None
        self._plan_fingerprint = None

    def _fingerprint(self):
        return tuple((t.data_ptr(), t._version) for t in list(self.parameters()) + list(self.buffers()))

    def _placement(self):
        p = next(self.parameters())
        return (p.device, p.dtype, p.data_ptr())

    def _apply(self, fn, *a, **k):
        before = self._placement()
        out = super()._apply(fn, *a, **k)
        if self._placement() != before:
            self._plan = None
        return out

    def load_state_dict(self, *a, **k):
        self._plan = None
        return super().load_state_dict(*a, **k)

    def invalidate_plan(self):
        """Drop the packed weights; needed after writes through ``p.data`` (see the module docstring)."""
        self._plan = None

    def refresh_plan_if_stale(self):
        if self._plan is not None and self._plan_fingerprint != self._fingerprint():
            self._plan = None

    def _stamp_plan(self):
        self._plan_fingerprint = self._fingerprint()
