"""
Object (de)serialisation with the reference's file conventions (mzd/io_utils.py:12-76): ``save_object``
always gzips and appends ``.gz`` when missing; ``load_object`` / ``open_input`` pick gzip, bz2 or plain by
file suffix.  Python-3 pickle replaces cPickle; maps pickled by the Python-2 reference load with
``encoding='latin1'``.
"""
import bz2
import gzip
import io
import pickle


def open_input(file_name):
    """Binary read handle, transparently decompressing ``.gz`` / ``.bz2`` (io_utils.py:35-49)."""
    suffix = file_name.rsplit('.', 1)[-1].lower()
    if suffix == 'bz2':
        return bz2.BZ2File(file_name, 'rb')
    if suffix == 'gz':
        return gzip.GzipFile(file_name, 'rb')
    return open(file_name, 'rb')


def open_output(file_name, append=False, compress=None, gzlevel=6):
    """Binary write handle; ``compress`` in (None, 'gzip', 'bzip2') adds the matching suffix if absent (io_utils.py:52-78)."""
    mode = 'ab' if append else 'wb'
    if compress == 'bzip2':
        if not file_name.endswith('.bz2'):
            file_name += '.bz2'
        return bz2.BZ2File(file_name, mode)
    if compress == 'gzip':
        if not file_name.endswith('.gz'):
            file_name += '.gz'
        return io.BufferedWriter(gzip.GzipFile(file_name, mode, compresslevel=gzlevel))
    return io.BufferedWriter(io.FileIO(file_name, mode))


def save_object(file_name, obj):
    """Pickle ``obj`` gzip-compressed to ``file_name`` (+ '.gz') -- io_utils.py:12-21."""
    with open_output(file_name, compress='gzip') as out_h:
        pickle.dump(obj, out_h, protocol=pickle.HIGHEST_PROTOCOL)


def load_object(file_name):
    """Inverse of save_object; also reads Python-2 pickles of plain data (io_utils.py:24-32)."""
    with open_input(file_name) as in_h:
        try:
            return pickle.load(in_h)
        except UnicodeDecodeError:
            in_h.seek(0)
            return pickle.load(in_h, encoding='latin1')
