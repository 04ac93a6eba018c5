"""Logger with the reference's extra VERBOSE level (reference _log.py:11-46)."""
import logging
import sys

VERBOSE = 5
logging.addLevelName(VERBOSE, "VERBOSE")


def _verbose(self, msg, *args, **kwargs):
    if self.isEnabledFor(VERBOSE):
        self._log(VERBOSE, msg, args, **kwargs)


logging.Logger.verbose = _verbose


def setup_logging(name, verbose=False):
    logger = logging.getLogger(name)
    logger.setLevel(logging.DEBUG if verbose else logging.WARNING)
    if not logger.handlers:
        h = logging.StreamHandler(sys.stderr)
        h.setFormatter(logging.Formatter("%(asctime)s - [PID %(process)d] - %(name)-25s - %(levelname)s - %(message)s"))
        logger.addHandler(h)
    return logger
