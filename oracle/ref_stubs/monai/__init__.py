"""Stand-in for `monai` (oracle-only): the one transform the reference's networks call lives in .transforms."""
from . import transforms  # noqa: F401
