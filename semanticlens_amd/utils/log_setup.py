"""Optional console / file logging for the package (reference: utils/log_setup.py:60-113).

``setup_colored_logging(level, file_path)`` puts one stream handler (ANSI-coloured when the stream is a terminal)
and, if asked, one plain file handler on the ``semanticlens_amd`` logger, replacing whatever handlers it had.  The
``SEMANTICLENS_LOG_LEVEL`` environment variable (same name as upstream) overrides the ``log_level`` argument.  Until it
is called the package logger only carries a ``NullHandler``.
"""
from __future__ import annotations

import logging
import os

PACKAGE = "semanticlens_amd"
_LINE = "[%(asctime)s|%(name)s|%(levelname)s]: %(message)s"


class ColorFormatter(logging.Formatter):
    """``logging.Formatter`` that wraps the formatted line in the level's ANSI colour (when ``use_color``) and gives
    records a ``short_filename`` attribute (basename of ``pathname``) for format strings that want it."""

    RESET_SEQ = "\033[0m"
    COLOR_MAP = {
        "DEBUG": "\033[90m",
        "INFO": "\033[92m",
        "WARNING": "\033[38;5;208m",
        "ERROR": "\033[91m",
        "CRITICAL": "\033[91m",
    }

    def __init__(self, fmt, use_color: bool = True):
        super().__init__(fmt)
        self.use_color = use_color

    def format(self, record):
        record.short_filename = os.path.basename(record.pathname)
        line = super().format(record)
        if not self.use_color:
            return line
        return self.COLOR_MAP.get(record.levelname, "") + line + self.RESET_SEQ


def _handler(handler: logging.Handler, level: int, color: bool) -> logging.Handler:
    handler.setLevel(level)
    handler.setFormatter(ColorFormatter(_LINE, use_color=color))
    return handler


def setup_colored_logging(log_level: str = "INFO", file_path: str | None = None):
    """Configure the package logger: level (``SEMANTICLENS_LOG_LEVEL`` wins over ``log_level``; unknown names mean
    INFO), a stream handler, and a file handler when ``file_path`` is given."""
    name = os.environ.get("SEMANTICLENS_LOG_LEVEL", log_level).upper()
    level = getattr(logging, name, logging.INFO)
    level = level if isinstance(level, int) else logging.INFO
    logger = logging.getLogger(PACKAGE)
    logger.setLevel(level)
    for old in list(logger.handlers):
        logger.removeHandler(old)
    stream = logging.StreamHandler()
    tty = bool(getattr(stream.stream, "isatty", lambda: False)())
    logger.addHandler(_handler(stream, level, tty))
    if file_path:
        logger.addHandler(_handler(logging.FileHandler(file_path), level, False))
    return logger


logging.getLogger(PACKAGE).addHandler(logging.NullHandler())
