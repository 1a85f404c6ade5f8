"""File lookup helpers of the reference's scripts (rl4rs/utils/fileutil.py:7-24): ``find_match_files`` yields the files
matching a glob pattern under each directory of an ``os.pathsep``-separated search path (script/supervised_train.py:33-34
lists the TFRecord shards with it); ``find_newest_files`` returns the most recently created match or ``''``
(script/modelfree_train.py:72-73 picks the latest checkpoint)."""
import glob
import os


def _matches(pattern, search_path, pathsep):
    for directory in search_path.split(pathsep):
        for path in glob.glob(os.path.join(directory, pattern)):
            yield path


def find_match_files(pattern, search_path, pathsep=os.pathsep):
    return _matches(pattern, search_path, pathsep)


def find_newest_files(pattern, search_path, pathsep=os.pathsep):
    best, best_time = '', None
    for path in _matches(pattern, search_path, pathsep):
        created = float(os.path.getctime(path))
        if best_time is None or created > best_time:        # first of equally new files wins, like np.argmax
            best, best_time = path, created
    return best
