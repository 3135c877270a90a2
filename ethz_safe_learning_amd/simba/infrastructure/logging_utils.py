"""Logger + scalar writer with the interface of reference simba/infrastructure/logging_utils.py:5-66.  tensorboardX is not
installed here, so scalars go to a JSON-lines file (one {"tag","value","step"} per line) under the log directory."""
import json
import logging
import os

logger = logging.getLogger('simba')
if not logger.handlers:
    _h = logging.StreamHandler()
    _h.setFormatter(logging.Formatter('%(asctime)s [%(levelname)s] %(message)s'))
    logger.addHandler(_h)
    logger.setLevel(logging.INFO)


def init_logging(log_level):
    logger.setLevel(getattr(logging, str(log_level).upper(), logging.INFO))


class TrainingLogger(object):
    def __init__(self, log_dir=None, fps=60, **_):
        self.log_dir = log_dir
        self.fps = fps
        self.scalars = []
        self._fh = None
        if log_dir:
            os.makedirs(log_dir, exist_ok=True)
            self._fh = open(os.path.join(log_dir, 'scalars.jsonl'), 'a')

    def log_scalar(self, scalar, name, step):
        rec = dict(tag=name, value=float(scalar), step=int(step))
        self.scalars.append(rec)
        if self._fh:
            self._fh.write(json.dumps(rec) + '\n')

    def log_video(self, *args, **kwargs):
        logger.debug('video logging is not available without a renderer')

    def flush(self):
        if self._fh:
            self._fh.flush()
