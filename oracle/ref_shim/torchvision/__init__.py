"""Test-infrastructure stub (NOT shipped, NOT used by the product path).

The reference (`/root/reference/model/propagation/modules.py:10,70`) imports
``torchvision.models.resnet50`` only to obtain the ResNet-50 v1.5 *architecture*;
torchvision itself is not installed in this image.  This stub provides just that
constructor so the unmodified reference can be imported on CPU when golden vectors
are generated (see oracle/make_golden.py).
"""
from . import models  # noqa: F401
from . import transforms  # noqa: F401  (test-time loaders: oracle/make_golden_dataset.py)
