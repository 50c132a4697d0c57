"""Drop-in for the reference's `utils.logger_utils` (reference: utils/logger_utils.py:5-48), which its
training scripts import next to the env (dqn_stable_baselines3.py:6, 12-14).  Same name, arguments
and behaviour: a named logger with an optional console handler and an optional `<log_dir>/<name>.log`
file handler, each with its own level and format; configured once per name."""
import logging
import os

_CONSOLE_FORMAT = '%(name)s - %(levelname)s - %(message)s'
_FILE_FORMAT = '%(asctime)s - %(name)s - %(levelname)s - %(message)s'


def setup_logger(name, log_dir, console_level=logging.INFO, file_level=logging.DEBUG, console_format_str=None,
                 file_format_str=None):
    """Return `logging.getLogger(name)`; on first use attach a console handler (unless `console_level` is
    None) and a file handler writing `<log_dir>/<name>.log` (unless `file_level` is None).  The directory is
    created when missing.  A logger that already has handlers is returned untouched."""
    os.makedirs(log_dir, exist_ok=True)
    logger = logging.getLogger(name)
    if logger.hasHandlers():
        return logger
    logger.setLevel(logging.DEBUG)                      # the handlers do the filtering
    wanted = []
    if console_level is not None:
        wanted.append((logging.StreamHandler(), console_level, console_format_str or _CONSOLE_FORMAT))
    if file_level is not None:
        wanted.append((logging.FileHandler(os.path.join(log_dir, f'{name}.log')), file_level,
                       file_format_str or _FILE_FORMAT))
    for handler, level, fmt in wanted:
        handler.setLevel(level)
        handler.setFormatter(logging.Formatter(fmt))
        logger.addHandler(handler)
    return logger
