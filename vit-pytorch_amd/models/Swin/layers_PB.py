"""The piggyback layers of ``/root/reference/models/Swin/layers_PB.py`` (identical to the reference's
``models/ViT/layers_PB.py``): one implementation serves both model families."""
from models.ViT.layers_PB import (DEFAULT_THRESHOLD, Binarizer, ElementWiseLinear, ElementWiseRG,  # noqa: F401
                                  ElementWiseSFG, Ternarizer)
