"""Small helpers with the reference's names and behaviour (mzd/utils.py:6-37)."""
import os
import sys

import numpy as np


def make_random_seed():
    """A seed in [1e6, 1e7) like the reference draws when -s is not given (utils.py:11, bin3C.py:169-173)."""
    return int(np.random.randint(1000000, 10000000))


def make_dir(path, exist_ok=False):
    """Create ``path``; an existing directory is an IOError unless exist_ok, an existing file always is (utils.py:14-26)."""
    if os.path.isfile(path):
        raise IOError('output path already exists and is a file!')
    if os.path.isdir(path):
        if not exist_ok:
            raise IOError('output directory already exists!')
        return
    os.mkdir(path)


def app_path(subdir, filename):
    """Path of a bundled helper relative to the running script's directory (utils.py:29-37)."""
    return os.path.join(sys.path[0], subdir, filename)
