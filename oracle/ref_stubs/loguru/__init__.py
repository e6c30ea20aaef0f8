"""Stand-in for `loguru` (oracle-only): routes to the stdlib logger."""
import logging

logger = logging.getLogger("ganslate-ref")
