"""Seeding / logging plumbing (ganslate/utils/environment.py:75-81)."""
import logging
import os
import random

import numpy as np
import torch

logger = logging.getLogger("ganslate_amd")


def set_seed(seed=0):
    logger.info(f"Reproducible mode ON with seed : {seed}")
    torch.manual_seed(seed)
    np.random.seed(seed)
    random.seed(seed)
    os.environ["PYTHONHASHSEED"] = str(seed)


def setup_logging(level=logging.INFO):
    if not logging.getLogger().handlers:
        logging.basicConfig(level=level, format="%(asctime)s | %(levelname)s | %(message)s")
