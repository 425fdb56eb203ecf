"""std logging logger, as mmgclip/utils/logger.py:11 (`logger = logging.getLogger(...)`)."""
import logging

logger = logging.getLogger("mmgclip")
if not logger.handlers:
    _h = logging.StreamHandler()
    _h.setFormatter(logging.Formatter("[%(asctime)s][%(levelname)s] %(message)s"))
    logger.addHandler(_h)
    logger.setLevel(logging.INFO)
    logger.propagate = False
