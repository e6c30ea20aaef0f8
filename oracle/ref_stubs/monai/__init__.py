"""Empty stand-in for `monai` (oracle-only)."""
