"""Empty stand-in for `torchvision` (oracle-only; data pipeline is out of scope)."""
from . import transforms
